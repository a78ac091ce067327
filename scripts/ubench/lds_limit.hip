#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out, int n) { extern __shared__ int s[]; for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = i; __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = s[n - 1]; }
int main() {
    int *d; hipMalloc(&d, 4096);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu sharedMemPerBlockOptin %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.maxSharedMemoryPerMultiProcessor);
    for (int kb : {48, 64, 96, 128, 160}) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        k<<<4, 256, kb * 1024>>>(d, kb * 256);
        hipError_t e2 = hipDeviceSynchronize(); hipError_t e3 = hipGetLastError();
        int h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%d KB: attr %s sync %s last %s out %d (want %d)\n", kb, hipGetErrorName(e), hipGetErrorName(e2), hipGetErrorName(e3), h, kb * 256 - 1);
    }
}
