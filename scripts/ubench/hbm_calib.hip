// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950: kernels that read / write a KNOWN
// number of bytes with the access widths libpolyhip's kernels use (4 B and 16 B per lane, coalesced).
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void read4(const uint32_t *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void read16(const uint4 *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void write4(uint32_t *p, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void write8(double *p, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (double)i;
}
__global__ __launch_bounds__(256) void write16(uint4 *p, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4(i, i, i, i);
}
int main()
{
    const size_t bytes = 2ull << 30; // 2 GiB: well past the 256 MiB Infinity Cache
    void *a, *b; uint32_t *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    const int g = 256 * 16;
    read4<<<g, 256>>>((const uint32_t *)a, bytes / 4, sink);
    read16<<<g, 256>>>((const uint4 *)b, bytes / 16, sink);
    write4<<<g, 256>>>((uint32_t *)a, bytes / 4);
    write8<<<g, 256>>>((double *)b, bytes / 8);
    write16<<<g, 256>>>((uint4 *)a, bytes / 16);
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
