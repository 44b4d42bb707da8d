// hipMemRelease cost vs order: create N 2-MiB handles, release them oldest-first or newest-first, time both.
//   hipcc --offload-arch=gfx950 -O3 tools/vmm_release_probe.cpp -o vmm_release_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(0);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t page = 2u << 20;
    for (int order = 0; order < 3; order++) {
        const int N = 40000;
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        double t0 = now();
        for (int i = 0; i < N; i++)
            if (hipMemCreate(&h[i], page, &prop, 0) != hipSuccess) { printf("create failed at %d\n", i); return 1; }
        double t1 = now();
        double first = 0, last = 0;
        for (int k = 0; k < N; k++) {
            int i = order == 0 ? k : order == 1 ? N - 1 - k : (k % 2 ? N - 1 - k / 2 : k / 2);
            double a = now();
            (void)hipMemRelease(h[i]);
            double b = now();
            if (k < 1000) first += b - a;
            if (k >= N - 1000) last += b - a;
        }
        double t2 = now();
        printf("%-22s create %d handles %.2f s | release %.2f s (first 1000: %.1f us each, last 1000: %.1f us each)\n",
               order == 0 ? "oldest-first" : order == 1 ? "newest-first" : "alternating ends", N, t1 - t0, t2 - t1, first * 1e3, last * 1e3);
    }
    return 0;
}
