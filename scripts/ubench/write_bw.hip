// write_bw.hip -- what a streaming WRITE sustains on MI355X, launch after launch (K4's bound: 24 B out per window).
//   hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip && ./write_bw
// Patterns: (1) K4's: a block of 256 lanes writes 39 planes x 2 KB (8 B per lane); (2) the same bytes as one linear
// stream, 16 B per lane; (3) 39 planes with 16 B per lane (two starts per lane); each timed as ONE launch (event pair
// around it) and as 20 back-to-back launches inside one event pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void planes8(double *out, uint64_t ld, int np, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = (double)i;
    for (int p = 0; p < np; ++p)
        out[(uint64_t)p * ld + i] = v + p;
}
__global__ __launch_bounds__(256) void planes16(double *out, uint64_t ld, int np, uint64_t n)
{
    const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const double v = (double)i;
    for (int p = 0; p < np; ++p)
        *reinterpret_cast<double2 *>(out + (uint64_t)p * ld + i) = make_double2(v + p, v - p);
}
__global__ __launch_bounds__(256) void linear16(double2 *out, uint64_t n2)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * 256)
        out[i] = make_double2((double)i, 1.0);
}
// planes, nontemporal stores
__global__ __launch_bounds__(256) void planes8_nt(double *out, uint64_t ld, int np, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = (double)i;
    for (int p = 0; p < np; ++p)
        __builtin_nontemporal_store(v + p, out + (uint64_t)p * ld + i);
}

// K4's round-4 ownership: a block of 256 lanes computes 256 consecutive elements but OWNS, in every plane, the 240 of them
// (15 lines) that start on that plane's own line boundary -- no 128-byte line is written from two blocks (two XCDs, two
// L2s); inside the block the waves still share lines (same CU, same L2)
__global__ __launch_bounds__(256) void planes8_own(double *out, uint64_t ld, int np, uint64_t n)
{
    const uint64_t T = (uint64_t)blockIdx.x * 240, col = T + threadIdx.x;
    const bool last = blockIdx.x == gridDim.x - 1;
    const double v = (double)col;
    for (int p = 0; p < np; ++p) {
        const uint64_t a = (reinterpret_cast<uintptr_t>(out + (uint64_t)p * ld) >> 3) & 15; // element phase of the plane's start
        const uint64_t r = (16 - a) & 15;                                                   // cols = r mod 16 start a line
        const uint64_t lo = blockIdx.x == 0 ? 0 : T + r, hi = last ? n : T + 240 + r;
        if (col >= lo && col < hi && col < n)
            out[(uint64_t)p * ld + col] = v + p;
    }
}

template <class F> void bench(const char *name, double bytes, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> one, grp;
    for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); one.push_back(ms);
    }
    for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(a)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); grp.push_back(ms / 20);
    }
    std::sort(one.begin(), one.end()); std::sort(grp.begin(), grp.end());
    printf("%-34s single %.4f ms = %6.0f GB/s   sustained %.4f ms = %6.0f GB/s\n", name, one[5], bytes / one[5] / 1e6, grp[5], bytes / grp[5] / 1e6);
}

int main()
{
    const uint64_t n = 4999983, ld = n;
    const int np = 39;
    double *out;
    CK(hipMalloc(&out, (size_t)np * ld * 8 + 4096));
    const double bytes = (double)np * n * 8;
    bench("39 planes, 8 B per lane", bytes, [&] { hipLaunchKernelGGL(planes8, dim3((n + 255) / 256), dim3(256), 0, 0, out, ld, np, n); });
    bench("39 planes, 8 B per lane, nt", bytes, [&] { hipLaunchKernelGGL(planes8_nt, dim3((n + 255) / 256), dim3(256), 0, 0, out, ld, np, n); });
    bench("39 planes, 16 B per lane", bytes, [&] { hipLaunchKernelGGL(planes16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, out, ld + 1, np, n); });
    bench("39 planes, 8 B, ld = 0 mod 16", bytes, [&] { hipLaunchKernelGGL(planes8, dim3((n + 255) / 256), dim3(256), 0, 0, out, ld + 1, np, n); });
    bench("39 planes, 8 B, owned lines, ld odd", bytes, [&] { hipLaunchKernelGGL(planes8_own, dim3((n + 239) / 240), dim3(256), 0, 0, out, ld, np, n); });
    bench("39 planes, 8 B, owned lines, ld = 2 mod 16", bytes, [&] { hipLaunchKernelGGL(planes8_own, dim3((n + 239) / 240), dim3(256), 0, 0, out, ld + 3, np, n); });
    bench("39 planes, 16 B, ld = 2 mod 16", bytes, [&] { hipLaunchKernelGGL(planes16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, out, ld + 3, np, n); });
    bench("39 planes, 16 B, ld = 8 mod 16", bytes, [&] { hipLaunchKernelGGL(planes16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, out, ld + 9, np, n); });
    bench("linear, 16 B per lane, 8192 blocks", bytes, [&] { hipLaunchKernelGGL(linear16, dim3(8192), dim3(256), 0, 0, (double2 *)out, (uint64_t)(bytes / 16)); });
    bench("13 planes, 8 B per lane", bytes / 3, [&] { hipLaunchKernelGGL(planes8, dim3((n + 255) / 256), dim3(256), 0, 0, out, ld, 13, n); });
    return 0;
}
