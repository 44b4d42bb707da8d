// Can one large physical handle be mapped piecewise (hipMemMap with size < handle size and/or offset != 0)?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CKP(x) do { hipError_t e_ = (x); printf("  %-90s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } while (0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  !! %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
__global__ void fill_k(unsigned* p, size_t n, unsigned v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void peek_k(const unsigned* p, unsigned* out) { if (threadIdx.x == 0) out[0] = p[0]; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    CK(hipSetDevice(0));
    hipMemAllocationProp ap = {}; ap.type = hipMemAllocationTypePinned; ap.location.type = hipMemLocationTypeDevice; ap.location.id = 0;
    hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
    unsigned* dout; CK(hipMalloc(&dout, 64));
    const size_t big = 64ul << 20, page = 65536;
    hipMemGenericAllocationHandle_t H;
    double t0 = now_us();
    CK(hipMemCreate(&H, big, &ap, 0));
    printf("create 64 MiB handle: %.1f us\n", now_us() - t0);
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, 4 * big, 2 << 20, nullptr, 0));
    // whole handle at va, stamp every 64 KiB piece with its index
    CK(hipMemMap(va, big, 0, H, 0)); CK(hipMemSetAccess(va, big, &ad, 1));
    for (unsigned i = 0; i < big / page; i += 97) { fill_k<<<4, 256>>>((unsigned*)(va + i * page), page / 4, 0xAB000000u + i); }
    CK(hipDeviceSynchronize());
    printf("partial map, offset 0, size 64 KiB:\n");
    CKP(hipMemMap(va + big, page, 0, H, 0));
    printf("map with offset 97 * 64 KiB, size 64 KiB:\n");
    hipError_t e = hipMemMap(va + 2 * big, page, 97 * page, H, 0);
    printf("  hipMemMap(offset=97*64K) -> %s\n", hipGetErrorString(e)); (void)hipGetLastError();
    if (e == hipSuccess) {
        CKP(hipMemSetAccess(va + 2 * big, page, &ad, 1));
        peek_k<<<1, 64>>>((const unsigned*)(va + 2 * big), dout);
        unsigned h = 0; CK(hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost));
        printf("  kernel reads %x through the offset mapping (expect ab000061)\n", h);
        // timing of many offset maps
        double a0 = now_us();
        int ok = 0;
        for (int i = 0; i < 256; i++) { if (hipMemMap(va + 3 * big + (size_t)i * page, page, (size_t)(2 * i) * page, H, 0) == hipSuccess) ok++; }
        double a1 = now_us();
        CK(hipMemSetAccess(va + 3 * big, 256 * page, &ad, 1));
        double a2 = now_us();
        printf("  256 offset maps: %d ok, %.2f us/map, merged access %.2f us/page\n", ok, (a1 - a0) / 256, (a2 - a1) / 256);
        peek_k<<<1, 64>>>((const unsigned*)(va + 3 * big + 97 * page), dout);     // piece 194 = 2*97
        CK(hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost));
        printf("  piece mapped from offset 194*64K reads %x (expect ab0000c2)\n", h);
        double u0 = now_us();
        for (int i = 0; i < 256; i++) CK(hipMemUnmap(va + 3 * big + (size_t)i * page, page));
        printf("  unmap %.2f us/call\n", (now_us() - u0) / 256);
    }
    return 0;
}
