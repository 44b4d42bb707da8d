// How do HIP VMM call costs scale with the NUMBER of live handles (small pages => hundreds of thousands of handles)?
// Build: hipcc --offload-arch=gfx950 -O2 tools/vmm_scale_probe.cpp -o tools/vmm_scale_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  !! %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t page = argc > 1 ? strtoul(argv[1], 0, 10) : 65536;
    const size_t maxn = argc > 2 ? strtoul(argv[2], 0, 10) : 200000;
    const double budget_ms = argc > 3 ? atof(argv[3]) : 40000.0;
    CK(hipSetDevice(0));
    hipMemAllocationProp ap = {}; ap.type = hipMemAllocationTypePinned; ap.location.type = hipMemLocationTypeDevice; ap.location.id = 0;
    hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, page * maxn, 2 << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h;
    h.reserve(maxn);
    const double start = now_ms();
    size_t mapped = 0;
    for (size_t target : {1000ul, 5000ul, 20000ul, 50000ul, 100000ul, 200000ul, 400000ul}) {
        if (target > maxn) break;
        double t0 = now_ms();
        while (h.size() < target) { hipMemGenericAllocationHandle_t x; CK(hipMemCreate(&x, page, &ap, 0)); h.push_back(x); }
        double t1 = now_ms();
        const size_t n_new = target - mapped;
        for (; mapped < target; mapped++) { CK(hipMemMap(va + mapped * page, page, 0, h[mapped], 0)); }
        double t2 = now_ms();
        CK(hipMemSetAccess(va + (target - n_new) * page, n_new * page, &ad, 1));
        double t3 = now_ms();
        // unmap + remap the last 200 pages
        for (size_t i = target - 200; i < target; i++) CK(hipMemUnmap(va + i * page, page));
        double t4 = now_ms();
        for (size_t i = target - 200; i < target; i++) { CK(hipMemMap(va + i * page, page, 0, h[i], 0)); CK(hipMemSetAccess(va + i * page, page, &ad, 1)); }
        double t5 = now_ms();
        printf("live handles %7zu: create %.2f us/call, map %.2f us/call, merged access %.2f us/page, unmap %.2f us/call, map+access %.2f us/page  (elapsed %.1f s)\n",
               target, (t1 - t0) * 1e3 / n_new, (t2 - t1) * 1e3 / n_new, (t3 - t2) * 1e3 / n_new, (t4 - t3) * 1e3 / 200, (t5 - t4) * 1e3 / 200, (t5 - start) / 1e3);
        fflush(stdout);
        if (now_ms() - start > budget_ms) { printf("time budget reached\n"); break; }
    }
    return 0;
}
