// duplex.hip -- do pageable uploads and downloads overlap when two host threads issue them?  (the host-pointer pipelines'
// Downloader thread, host_pipeline.h)   hipcc --offload-arch=gfx950 -O2 -o duplex duplex.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 64, n = mb << 20;
    const int reps = 16;
    char *hu = (char *)malloc(n * reps), *hd = (char *)malloc(n * reps);
    memset(hu, 1, n * reps); memset(hd, 2, n * reps);
    char *du, *dd;
    CK(hipMalloc(&du, n)); CK(hipMalloc(&dd, n));
    hipStream_t su, sd;
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    auto up = [&] { for (int r = 0; r < reps; ++r) { CK(hipMemcpyAsync(du, hu + r * n, n, hipMemcpyHostToDevice, su)); } CK(hipStreamSynchronize(su)); };
    auto down = [&] { CK(hipSetDevice(0)); for (int r = 0; r < reps; ++r) { CK(hipMemcpyAsync(hd + r * n, dd, n, hipMemcpyDeviceToHost, sd)); } CK(hipStreamSynchronize(sd)); };
    up(); down();
    double t = now(); up(); const double tu = now() - t;
    t = now(); down(); const double td = now() - t;
    t = now(); { std::thread th(down); up(); th.join(); } const double tb = now() - t;
    const double gb = (double)n * reps / 1e9;
    printf("%zu MB chunks x %d: up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s), both from two threads %.2f ms (%.1f GB/s total)\n", mb, reps, tu,
           gb / tu * 1e3, td, gb / td * 1e3, tb, 2 * gb / tb * 1e3);
    return 0;
}
