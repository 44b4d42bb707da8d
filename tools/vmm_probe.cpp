// HIP virtual-memory probe for gfx950: the numbers DESIGN.md quotes for the page manager's backend.
//   granularity (min / recommended), legal page sizes, per-call latency of hipMemCreate / hipMemMap /
//   hipMemSetAccess (per page and per merged run) / hipMemUnmap, size of the largest VA reservation,
//   aliasing one handle at two addresses (map_common_pages), mapping while a kernel runs.
// Build: hipcc --offload-arch=gfx950 -O2 tools/vmm_probe.cpp -o tools/vmm_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("  !! %s -> %s\n", #x, hipGetErrorString(e_));                   \
            (void)hipGetLastError();                                               \
        }                                                                          \
    } while (0)

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void spin_kernel(float* p, long long cycles) {
    const long long t0 = wall_clock64();                 // 100 MHz constant clock
    long long n = 0;
    while (wall_clock64() - t0 < cycles) n++;
    p[threadIdx.x] = (float)n;
}
__global__ void touch_kernel(char* p, size_t n, char v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("device: %s  arch %s  CUs %d  HBM total %.1f GiB free %.1f GiB\n", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, total_b / 1073741824.0, free_b / 1073741824.0);
    hipMemAllocationProp ap = {};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    ap.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &ap, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &ap, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu  recommended %zu\n", gmin, grec);
    hipMemAccessDesc ad = {};
    ad.location.type = hipMemLocationTypeDevice;
    ad.location.id = 0;
    ad.flags = hipMemAccessFlagsProtReadWrite;

    // largest VA reservation
    for (size_t gib : {64ul, 512ul, 1024ul, 4096ul, 16384ul}) {
        void* p = nullptr;
        double t0 = now_us();
        hipError_t e = hipMemAddressReserve(&p, gib << 30, 2 << 20, nullptr, 0);
        double t1 = now_us();
        printf("reserve %6zu GiB: %s (%.1f us)\n", gib, e == hipSuccess ? "ok" : hipGetErrorString(e), t1 - t0);
        if (e == hipSuccess) CK(hipMemAddressFree(p, gib << 30)); else (void)hipGetLastError();
    }

    for (size_t page : {4096ul, 65536ul, 262144ul, 2097152ul}) {
        printf("---- page size %zu ----\n", page);
        if (page % gmin) { printf("  not a multiple of the minimum granularity: skipped\n"); continue; }
        const int N = 256;
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        double t0 = now_us();
        int created = 0;
        for (int i = 0; i < N; i++) {
            hipError_t e = hipMemCreate(&h[i], page, &ap, 0);
            if (e != hipSuccess) { printf("  hipMemCreate failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
            created++;
        }
        double t1 = now_us();
        printf("  hipMemCreate        %8.2f us/call (%d ok)\n", (t1 - t0) / (created ? created : 1), created);
        if (created < N) { for (int i = 0; i < created; i++) CK(hipMemRelease(h[i])); continue; }
        char* va = nullptr;
        CK(hipMemAddressReserve((void**)&va, page * N * 2, page > grec ? page : grec, nullptr, 0));
        t0 = now_us();
        for (int i = 0; i < N; i++) CK(hipMemMap(va + i * page, page, 0, h[i], 0));
        t1 = now_us();
        printf("  hipMemMap           %8.2f us/call\n", (t1 - t0) / N);
        // per-page set access on the first half, one merged call on the second half
        t0 = now_us();
        for (int i = 0; i < N / 2; i++) CK(hipMemSetAccess(va + i * page, page, &ad, 1));
        t1 = now_us();
        printf("  hipMemSetAccess     %8.2f us/call (per page)\n", (t1 - t0) / (N / 2));
        t0 = now_us();
        hipError_t em = hipMemSetAccess(va + (N / 2) * page, page * (N / 2), &ad, 1);
        t1 = now_us();
        printf("  hipMemSetAccess     %8.2f us for ONE call over %d pages: %s\n", t1 - t0, N / 2, em == hipSuccess ? "ok" : hipGetErrorString(em));
        if (em != hipSuccess) { (void)hipGetLastError(); for (int i = N / 2; i < N; i++) CK(hipMemSetAccess(va + i * page, page, &ad, 1)); }
        touch_kernel<<<256, 256>>>(va, page * N, 7);
        CK(hipDeviceSynchronize());
        char probe[2] = {0, 0};
        CK(hipMemcpy(&probe[0], va, 1, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&probe[1], va + page * N - 1, 1, hipMemcpyDeviceToHost));
        printf("  touch through the mapping: first=%d last=%d (expect 7 7)\n", probe[0], probe[1]);
        // alias: same handle at a second address (map_common_pages)
        hipError_t ea = hipMemMap(va + N * page, page, 0, h[0], 0);
        printf("  alias same handle at 2nd VA: %s\n", ea == hipSuccess ? "ok" : hipGetErrorString(ea));
        if (ea == hipSuccess) {
            CK(hipMemSetAccess(va + N * page, page, &ad, 1));
            char c = 0;
            CK(hipMemcpy(&c, va + N * page, 1, hipMemcpyDeviceToHost));
            printf("    read through alias: %d (expect 7)\n", c);
            CK(hipMemUnmap(va + N * page, page));
        } else (void)hipGetLastError();
        // Which VMM calls block on running kernels?  For each call type, with a 0.3 s kernel in flight on
        // (a) a default-flag stream and (b) a hipStreamNonBlocking stream, time the call and ask whether
        // the kernel is still running afterwards.
        if (page == 2097152 || page == 65536) {
            float* buf;
            CK(hipMalloc(&buf, 1024 * sizeof(float)));
            for (int nb = 0; nb < 2; nb++) {
                hipStream_t s;
                CK(hipStreamCreateWithFlags(&s, nb ? hipStreamNonBlocking : hipStreamDefault));
                const char* names[4] = {"hipMemCreate", "hipMemMap", "hipMemSetAccess", "hipMemUnmap"};
                hipMemGenericAllocationHandle_t hx;
                char* tgt = va + (N + 8) * page;
                for (int op = 0; op < 4; op++) {
                    spin_kernel<<<1, 64, 0, s>>>(buf, 30000000LL);      // 0.3 s at 100 MHz
                    double a0 = now_us();
                    if (op == 0) CK(hipMemCreate(&hx, page, &ap, 0));
                    if (op == 1) CK(hipMemMap(tgt, page, 0, hx, 0));
                    if (op == 2) CK(hipMemSetAccess(tgt, page, &ad, 1));
                    if (op == 3) CK(hipMemUnmap(tgt, page));
                    double a1 = now_us();
                    hipError_t q = hipStreamQuery(s);
                    (void)hipGetLastError();
                    CK(hipStreamSynchronize(s));
                    printf("  [%s stream] %-16s %10.2f us, kernel still running afterwards: %s\n", nb ? "non-blocking" : "default-flag", names[op], a1 - a0,
                           q == hipErrorNotReady ? "yes" : "NO (call waited for it)");
                }
                CK(hipMemRelease(hx));
                CK(hipStreamDestroy(s));
            }
            CK(hipFree(buf));
        }
        t0 = now_us();
        for (int i = 0; i < N; i++) CK(hipMemUnmap(va + i * page, page));
        t1 = now_us();
        printf("  hipMemUnmap         %8.2f us/call\n", (t1 - t0) / N);
        // remap + one merged unmap
        for (int i = 0; i < 16; i++) CK(hipMemMap(va + i * page, page, 0, h[i], 0));
        t0 = now_us();
        hipError_t eu = hipMemUnmap(va, page * 16);
        t1 = now_us();
        printf("  hipMemUnmap over 16 pages in ONE call: %s (%.2f us)\n", eu == hipSuccess ? "ok" : hipGetErrorString(eu), t1 - t0);
        if (eu != hipSuccess) { (void)hipGetLastError(); for (int i = 0; i < 16; i++) CK(hipMemUnmap(va + i * page, page)); }
        t0 = now_us();
        for (int i = 0; i < N; i++) CK(hipMemRelease(h[i]));
        t1 = now_us();
        printf("  hipMemRelease       %8.2f us/call\n", (t1 - t0) / N);
        CK(hipMemAddressFree(va, page * N * 2));
    }
    return 0;
}
