// runtime.hip -- error plumbing, device queries and the synthetic-input
// generator of libpolyhip.so.
#include <cstring>

#include "common.h"

namespace polyhip {

static thread_local char g_err[512] = "";

int set_error(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

void clear_error() { g_err[0] = 0; }

// splitmix64 output number `idx` (1-based) of the stream seeded with `seed`
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// One thread per 32 bases (one 64-bit draw): two 16-byte stores per lane.
__global__ __launch_bounds__(256) void synth_dna_kernel(uint64_t seed, uint64_t first_word,
                                                        uint8_t *__restrict__ out, uint64_t n)
{
    const uint64_t nwords = (n + 31) / 32;
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < nwords;
         w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = splitmix64_at(seed, first_word + w + 1);
        uint32_t b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t code = (uint32_t)(x >> (2 * (4 * q + j))) & 3u;
                // "ACGT" = 0x41 0x43 0x47 0x54
                uint32_t ch = (0x54474341u >> (8 * code)) & 0xFFu;
                v |= ch << (8 * j);
            }
            b[q] = v;
        }
        const uint64_t base = w * 32;
        if (base + 32 <= n && ((uintptr_t)(out + base) & 15) == 0) {
            uint4 *o = reinterpret_cast<uint4 *>(out + base);
            o[0] = make_uint4(b[0], b[1], b[2], b[3]);
            o[1] = make_uint4(b[4], b[5], b[6], b[7]);
        } else {
            for (uint64_t i = 0; i < 32 && base + i < n; ++i)
                out[base + i] = (uint8_t)(b[i / 4] >> (8 * (i & 3)));
        }
    }
}

} // namespace polyhip

using namespace polyhip;

extern "C" {

int polyhip_abi_version(void) { return POLYHIP_ABI_VERSION; }

const char *polyhip_last_error(void) { return g_err; }

int polyhip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice)
        return 0;
    if (e != hipSuccess)
        return set_error(POLYHIP_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int polyhip_set_device(int device)
{
    PH_HIP(hipSetDevice(device));
    return POLYHIP_OK;
}

int polyhip_device_arch(char *buf, size_t buflen)
{
    PH_REQUIRE(buf && buflen > 0, "polyhip_device_arch: null buffer");
    int dev = 0;
    PH_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PH_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return POLYHIP_OK;
}

int polyhip_synth_dna_dev(uint64_t seed, uint64_t first, uint8_t *d_out, uint64_t n,
                          polyhip_stream_t stream)
{
    PH_REQUIRE(d_out || n == 0, "polyhip_synth_dna_dev: null output");
    PH_REQUIRE((first & 31) == 0, "polyhip_synth_dna_dev: first must be a multiple of 32");
    if (n == 0)
        return POLYHIP_OK;
    const uint64_t nwords = (n + 31) / 32;
    uint64_t blocks = (nwords + 255) / 256;
    if (blocks > 256 * 32)
        blocks = 256 * 32;
    hipLaunchKernelGGL(synth_dna_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       seed, first / 32, d_out, n);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

} // extern "C"
