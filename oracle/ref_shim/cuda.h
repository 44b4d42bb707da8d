// Fake CUDA driver ("fakecuda") for building /root/reference/vattention/vattention.cu
// as a host-only Python module.  TEST INFRASTRUCTURE ONLY (oracle/_ref): it lets the
// reference's own page-bookkeeping code run here so that oracle/pagemgr.py and the
// product's C++ manager can be pinned against it.  Every driver entry point the
// reference calls (vattention/cudaInternal.h:15-94, vtensor.h:21-46, uvmInternal.h:86-266)
// is replaced by a host function that records the call in an in-memory log.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cassert>
#include <stdexcept>
#include <dirent.h>
#include <unistd.h>
#include <sys/ioctl.h>
#include <sys/types.h>

typedef int CUresult;
#define CUDA_SUCCESS 0
typedef unsigned long long CUdeviceptr;
typedef void* CUcontext;
typedef unsigned long long CUmemGenericAllocationHandle;
enum { CU_MEM_ALLOCATION_TYPE_PINNED = 1 };
enum { CU_MEM_LOCATION_TYPE_DEVICE = 1 };
enum { CU_MEM_ACCESS_FLAGS_PROT_READWRITE = 3 };
enum { CU_MEM_ALLOC_GRANULARITY_MINIMUM = 0 };
struct CUmemLocation { int type; int id; };
struct CUmemAllocationProp { int type; int requestedHandleTypes; CUmemLocation location; };
struct CUmemAccessDesc { CUmemLocation location; int flags; };

extern "C" {
// log record kinds: 1 reserve, 2 create, 3 map, 4 setaccess, 5 unmap, 6 addrfree, 7 release,
//                   8 uvm_get_page, 9 uvm_map, 10 uvm_clear, 11 uvm_free_page
void fakecuda_log(int kind, unsigned long long a, unsigned long long b, unsigned long long c);
unsigned long long fakecuda_next_handle();
unsigned long long fakecuda_reserve(unsigned long long size);
}

static inline CUresult cuInit(unsigned) { return CUDA_SUCCESS; }
static inline CUresult cuCtxGetCurrent(CUcontext* c) { *c = (CUcontext)0x1; return CUDA_SUCCESS; }
static inline CUresult cuGetErrorString(CUresult, const char** s) { *s = "fakecuda"; return CUDA_SUCCESS; }
static inline CUresult cuMemGetAllocationGranularity(unsigned long* g, const CUmemAllocationProp*, int) {
    *g = 2UL * 1024 * 1024; return CUDA_SUCCESS; }
static inline CUresult cuMemAddressReserve(CUdeviceptr* p, size_t size, size_t, CUdeviceptr, unsigned long long) {
    *p = fakecuda_reserve(size); fakecuda_log(1, *p, size, 0); return CUDA_SUCCESS; }
static inline CUresult cuMemCreate(CUmemGenericAllocationHandle* h, size_t size, const CUmemAllocationProp*, unsigned long long) {
    *h = fakecuda_next_handle(); fakecuda_log(2, *h, size, 0); return CUDA_SUCCESS; }
static inline CUresult cuMemMap(CUdeviceptr p, size_t size, size_t off, CUmemGenericAllocationHandle h, unsigned long long) {
    fakecuda_log(3, p, size, h); (void)off; return CUDA_SUCCESS; }
static inline CUresult cuMemSetAccess(CUdeviceptr p, size_t size, const CUmemAccessDesc*, size_t) {
    fakecuda_log(4, p, size, 0); return CUDA_SUCCESS; }
static inline CUresult cuMemUnmap(CUdeviceptr p, size_t size) { fakecuda_log(5, p, size, 0); return CUDA_SUCCESS; }
static inline CUresult cuMemAddressFree(CUdeviceptr p, size_t size) { fakecuda_log(6, p, size, 0); return CUDA_SUCCESS; }
static inline CUresult cuMemRelease(CUmemGenericAllocationHandle h) { fakecuda_log(7, h, 0, 0); return CUDA_SUCCESS; }

// ---- runtime API bits used by vtensor.h / uvmInternal.h ----
typedef int cudaError_t;
struct cudaUUID_t { char bytes[16]; };
struct cudaDeviceProp { cudaUUID_t uuid; };
static inline cudaError_t cudaSetDevice(int) { return 0; }
template <typename T> static inline cudaError_t cudaMallocManaged(T** p, size_t size) {
    *p = (T*)fakecuda_reserve(size); return 0; }
static inline cudaError_t cudaFree(void*) { return 0; }
static inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int) { memset(p, 0, sizeof(*p)); return 0; }

// ---- c10::cuda bits used by vtensor.h ----
#include <c10/core/Device.h>
namespace c10 { namespace cuda {
static inline int GetDevice(c10::DeviceIndex* d) { *d = 0; return 0; }
}}
#define C10_CUDA_CHECK(x) do { (void)(x); } while (0)
#ifndef TORCH_CUDA_CPP_API
#define TORCH_CUDA_CPP_API
#endif
