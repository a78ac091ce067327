#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
    hipFree(0);
    size_t sizes[] = {8ull << 30, 525ull << 20, 525ull << 20, 150ull << 20, 64ull << 20, 8 << 20, 4 << 20, 1 << 20, 4096};
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        void *p[9];
        for (int i = 0; i < 9; ++i)
            hipMalloc(&p[i], sizes[i]);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 9; ++i)
            hipFree(p[i]);
        auto t2 = std::chrono::steady_clock::now();
        printf("malloc %.3f ms free %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        void *p;
        for (int i = 0; i < 10; ++i) { hipMalloc(&p, 4096); hipFree(p); }
        auto t1 = std::chrono::steady_clock::now();
        printf("10 x (malloc 4K + free): %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    return 0;
}
