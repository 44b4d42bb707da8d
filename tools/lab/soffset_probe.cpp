// LAB probe (round 6): is the SCALAR offset of a raw-buffer load part of the range check?  Lane l reads 16 bytes at voffset = 16 l with
// soffset = S from a descriptor of num_records = N bytes; a lane whose (voffset + S) lies at or beyond N must return zeros if the scalar
// offset is checked (then the per-piece deltas of the LDS-DMA stream can move from a v_mad per piece into a free scalar operand).
// The memory behind the bound is MAPPED (one allocation), so the probe cannot fault either way.
// build: hipcc --offload-arch=gfx950 -O2 tools/lab/soffset_probe.cpp -o tools/lab/soffset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out, unsigned nrec, unsigned soff) {
    const unsigned l = threadIdx.x;
    const unsigned long long a = (unsigned long long)src;
    u4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    r[2] = __builtin_amdgcn_readfirstlane(nrec);
    r[3] = 0x00020000u;
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    u4 v;
    const unsigned vo = l * 16;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(vo), "s"(r), "s"(so) : "memory");
    out[l] = v[0];
}
int main() {
    unsigned *src, *out, h[64], hs[4096];
    (void)hipMalloc(&src, 16384);
    (void)hipMalloc(&out, 256);
    for (int i = 0; i < 4096; i++) hs[i] = 0x1000 + i;      // word i holds 0x1000 + i: never zero
    (void)hipMemcpy(src, hs, 16384, hipMemcpyHostToDevice);
    const unsigned cases[][2] = {{1024, 0}, {1024, 512}, {1024, 1008}, {1024, 1024}, {512, 256}, {4096, 3584}};
    for (auto& c : cases) {
        k<<<1, 64>>>(src, out, c[0], c[1]);
        (void)hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        int first_zero = -1, wrong = 0;
        for (int l = 0; l < 64; l++) {
            const unsigned want_in = 0x1000 + (c[1] + l * 16) / 4;
            if (h[l] == 0 && first_zero < 0) first_zero = l;
            if (h[l] != 0 && h[l] != want_in) wrong++;
        }
        const int expect = (int)((c[0] - c[1] + 15) / 16);      // first lane with voffset + soffset >= num_records
        printf("num_records %5u soffset %5u: first zero lane %3d (soffset checked: lane %d; not checked: lane %d), wrong values %d\n", c[0], c[1], first_zero,
               expect > 63 ? -1 : expect, (int)(c[0] / 16) > 63 ? -1 : (int)(c[0] / 16), wrong);
    }
    return 0;
}
