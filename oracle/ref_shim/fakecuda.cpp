// Fake CUDA driver state + call log for the oracle/_ref build (see cuda.h). TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>

namespace {
struct Rec { int kind; unsigned long long a, b, c; };
std::vector<Rec> g_log;
std::mutex g_mu;
unsigned long long g_next_handle = 1000;
unsigned long long g_next_va = 0x700000000000ULL;
bool g_log_enabled = true;
}

extern "C" {
void fakecuda_log(int kind, unsigned long long a, unsigned long long b, unsigned long long c) {
    std::lock_guard<std::mutex> l(g_mu);
    if (g_log_enabled) g_log.push_back({kind, a, b, c});
}
unsigned long long fakecuda_next_handle() { std::lock_guard<std::mutex> l(g_mu); return g_next_handle++; }
unsigned long long fakecuda_reserve(unsigned long long size) {
    std::lock_guard<std::mutex> l(g_mu);
    const unsigned long long al = 2ULL << 20;
    unsigned long long p = g_next_va;
    g_next_va += ((size + al - 1) / al) * al + al;
    return p;
}
void fakecuda_reset() { std::lock_guard<std::mutex> l(g_mu); g_log.clear(); g_next_handle = 1000; g_next_va = 0x700000000000ULL; }
void fakecuda_enable_log(int on) { g_log_enabled = on != 0; }
long fakecuda_log_size() { std::lock_guard<std::mutex> l(g_mu); return (long)g_log.size(); }
long fakecuda_log_read(long start, long n, unsigned long long* out) {
    std::lock_guard<std::mutex> l(g_mu);
    long k = 0;
    for (long i = start; i < (long)g_log.size() && k < n; i++, k++) {
        out[4*k] = g_log[i].kind; out[4*k+1] = g_log[i].a; out[4*k+2] = g_log[i].b; out[4*k+3] = g_log[i].c;
    }
    return k;
}
void fakecuda_log_clear() { std::lock_guard<std::mutex> l(g_mu); g_log.clear(); }

char* fakecuda_realpath(const char* path, char*) {
    if (strncmp(path, "/proc/self/fd/", 14) == 0) return strdup("/dev/nvidia-uvm");
    return nullptr;
}
// ioctl numbers: uvmInternal.h:11-14.  Parameter layouts: uvmInternal.h:42-71.
int fakecuda_ioctl(int, unsigned long req, void* arg) {
    unsigned char* p = (unsigned char*)arg;
    unsigned long long v[3];
    switch (req) {
    case 200: { // get_mem_page {size, uuid[16], page, status}
        unsigned long long h = fakecuda_next_handle();
        memcpy(p + 24, &h, 8); memcpy(&v[0], p, 8); fakecuda_log(8, h, v[0], 0); return 0; }
    case 201: { // mem_map {base_address, page, uuid[16], size, status}
        memcpy(&v[0], p, 8); memcpy(&v[1], p + 8, 8); memcpy(&v[2], p + 32, 8);
        fakecuda_log(9, v[0], v[2], v[1]); return 0; }
    case 203: { // clear_address {base_address, size, uuid[16], status}
        memcpy(&v[0], p, 8); memcpy(&v[1], p + 8, 8); fakecuda_log(10, v[0], v[1], 0); return 0; }
    case 202: { // free_mem_page {page, size, uuid[16], status}
        memcpy(&v[0], p, 8); fakecuda_log(11, v[0], 0, 0); return 0; }
    }
    return -1;
}
}
