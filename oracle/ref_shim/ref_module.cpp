// Build wrapper for oracle/_ref: compiles the REFERENCE allocator *from where it lies*
// (/root/reference/vattention/vattention.cu, included below — never copied) against the
// fake CUDA driver in this directory, so its page-bookkeeping code runs on a CPU-only box.
// TEST INFRASTRUCTURE ONLY. Output goes to oracle/_ref/ (git-ignored).
//
// Everything the reference includes is pre-included here so that the four macro
// substitutions below apply to the reference's own code only, not to library headers.
#include <torch/extension.h>
#include <torch/types.h>
#include <ATen/ATen.h>
#include <ATen/ScalarType.h>
#include <ATen/ArrayRef.h>
#include <ATen/Tensor.h>
#include <c10/core/DeviceGuard.h>
#include <c10/util/static_tracepoint.h>
#include <vector>
#include <map>
#include <tuple>
#include <thread>
#include <atomic>
#include <chrono>
#include <iostream>
#include <iomanip>
#include <sstream>
#include <utility>
#include <Python.h>
#include "cuda.h"
#include "cuda_runtime.h"

// (1) at::globalContext().lazyInitCUDA() (vtensor.h:114) would throw without a GPU; the
//     tensors are never dispatched on, so route it to a harmless const query.
#define lazyInitCUDA() hasCUDA()
// (2) the small-page path discovers /dev/nvidia-uvm through /proc/self/fd and talks to the
//     patched driver through ioctl (uvmInternal.h:86-217): both are faked.
extern "C" char* fakecuda_realpath(const char* path, char* resolved);
extern "C" int fakecuda_ioctl(int fd, unsigned long req, void* arg);
#define realpath(p, r) fakecuda_realpath((p), (r))
#define ioctl(fd, req, arg) fakecuda_ioctl((fd), (req), (void*)(arg))
// (3) the background mapper is a detached thread (vattention.cu:538-547) that the caller
//     cannot join deterministically; running it to completion inside step_async gives the
//     "bg work completes before the next API call" semantics every real call sequence has
//     (the next call comes after a whole forward pass) and makes golden traces reproducible.
#define detach() join()
// (4) c10::DeviceGuard on a cuda device (vtensor.h:117) would call into the GPU runtime.
namespace c10 { struct RefNoopDeviceGuard { explicit RefNoopDeviceGuard(c10::Device) {} }; }
#define DeviceGuard RefNoopDeviceGuard

#include "vattention.cu"   // resolved through -I/root/reference/vattention

#undef DeviceGuard
#undef detach
#undef ioctl
#undef realpath
#undef lazyInitCUDA

// ---------------- introspection (ctypes) over the reference's file-scope state ----------------
extern "C" {
// layout: [B, tokens_per_page, max_pages_per_req, virt_per_req, virt_total, pool_size, pagemap_size,
//          mapped[0..B), lens[0..B), pool handles (bottom..top)]
long ref_dump_state(unsigned long long* out, long cap) {
    const bool uvm = is_uvm_backend(page_size);
    const long B = (long)mapped_pages.size();
    const long pool = uvm ? (long)uvm_pages.size() : (long)cuda_pages.size();
    const long need = 7 + 2 * B + pool;
    if (cap < need) return -need;
    long k = 0;
    out[k++] = B; out[k++] = tokens_per_page; out[k++] = max_pages_per_req;
    out[k++] = virt_buff_size_per_req; out[k++] = virt_buff_size; out[k++] = pool;
    out[k++] = uvm ? uvm_pagemap.size() : cuda_pagemap.size();
    for (long i = 0; i < B; i++) out[k++] = mapped_pages[i];
    for (long i = 0; i < B; i++) out[k++] = curr_seq_lengths[i];
    for (long i = 0; i < pool; i++) out[k++] = uvm ? uvm_pages[i] : cuda_pages[i];
    return k;
}
// pagemap entries as (reqId, offset, layer, kpage, vpage) rows, map order
long ref_dump_pagemap(unsigned long long* out, long cap_rows) {
    long n = 0;
    if (is_uvm_backend(page_size)) {
        for (auto& kv : uvm_pagemap) { if (n >= cap_rows) return -1;
            out[5*n] = std::get<0>(kv.first); out[5*n+1] = std::get<1>(kv.first); out[5*n+2] = std::get<2>(kv.first);
            out[5*n+3] = kv.second.first; out[5*n+4] = kv.second.second; n++; }
    } else {
        for (auto& kv : cuda_pagemap) { if (n >= cap_rows) return -1;
            out[5*n] = std::get<0>(kv.first); out[5*n+1] = std::get<1>(kv.first); out[5*n+2] = std::get<2>(kv.first);
            out[5*n+3] = kv.second.first; out[5*n+4] = kv.second.second; n++; }
    }
    return n;
}
// reset the reference's globals so several configurations can run in one process
void ref_reset() {
    cuda_pages.clear(); uvm_pages.clear(); cuda_pagemap.clear(); uvm_pagemap.clear();
    k_tensors.clear(); v_tensors.clear(); k_ptr.clear(); v_ptr.clear();
    mapped_pages.clear(); curr_seq_lengths.clear(); deferred_reclaim = true; verbose = false;
    page_size = 2 * MB; nvidia_uvm_fd = -1;
}
}
