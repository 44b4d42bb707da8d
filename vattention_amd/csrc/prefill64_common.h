// Building blocks shared by the two prefill64 kernels (prefill64_kernels.hip: one workgroup per query block; prefill64p_kernels.hip:
// persistent workgroups that walk a queue of query-block pieces): LDS-DMA pieces, the running buffer descriptors in fixed SGPR quads,
// and the matrix instructions with the register file of every operand fixed by the constraint.
#pragma once
#include "attn_common.h"

namespace vattn_k {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kDeferLog2 = 6.0f;      // P stays below 2^6 between rescales (f16 / bf16 keep their relative precision)

// ---- LDS-DMA: one 1-KiB piece (64 lanes x 16 bytes) of a K or V tile per call ----
// lds_addr: wave-uniform LDS byte address of the piece; rsrc: buffer descriptor of the tile, bounded at the visible rows; voff:
// per-lane byte offset inside the tile.  A lane whose offset lies beyond the descriptor fetches nothing (zeros).  s_nop 0 after
// the M0 write is the M0 -> LDS-DMA hazard; the _first form opens with s_nop 4 for a descriptor whose SGPRs were written by a
// VALU readfirstlane (cdna guide §5.7).  The compiler's wait-count pass does not see these loads: waits are counted by hand.
__device__ __forceinline__ void dma_piece(unsigned lds_addr, u32x4 rsrc, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds" : : "s"(lds_addr), "s"(rsrc), "v"(voff) : "memory", "m0");
}
__device__ __forceinline__ void dma_piece_first(unsigned lds_addr, u32x4 rsrc, unsigned voff) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds" : : "s"(lds_addr), "s"(rsrc), "v"(voff) : "memory", "m0");
}
// the same with the piece's LDS address formed in M0 by one SALU add (base register + literal): steady-state form, the descriptor's
// SGPRs are SALU-written (no VALU -> SGPR -> VMEM hazard, no s_nop 4)
template <int OFF> __device__ __forceinline__ void dma_piece_at(unsigned lds_base, u32x4 rsrc, unsigned voff) {
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds" : : "s"(lds_base), "s"(rsrc), "v"(voff), "i"(OFF) : "memory", "m0", "scc");
}
// the same with the piece's distance from piece 0 in the SCALAR offset operand (ROWS x row_bytes, formed by one s_mul in the wait state
// behind the M0 write): every piece of a tile then takes piece 0's per-lane offset — no v_mad per piece (a VALU instruction beside the
// MFMAs costs 7.3 cycles of this wave, a scalar one 0.3: profiles/r06_p64_price_list.txt).  The scalar offset is part of the descriptor's
// range check on gfx950 (tools/lab/soffset_probe.cpp: a lane with voffset + soffset >= num_records reads zeros), so rows beyond the
// sequence's visible length stay unfetched exactly as before.
template <int OFF, int ROWS> __device__ __forceinline__ void dma_piece_so(unsigned lds_base, u32x4 rsrc, unsigned voff, unsigned row_bytes) {
    unsigned so;
    asm volatile("s_add_u32 m0, %1, %6\n\ts_mul_i32 %0, %4, %5\n\tbuffer_load_dwordx4 %3, %2, %0 offen lds"
                 : "=&s"(so) : "s"(lds_base), "s"(rsrc), "v"(voff), "s"(row_bytes), "n"(ROWS), "i"(OFF) : "memory", "m0", "scc");
}
// base + ROWS x row_bytes per lane (row strides are far below 2^24 bytes)
template <int ROWS> __device__ __forceinline__ unsigned piece_off(unsigned base, unsigned row_bytes) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %3, %2" : "=v"(r) : "s"(row_bytes), "v"(base), "n"(ROWS));
    return r;
}
// The two running buffer descriptors of the steady state live in FIXED scalar registers — K(t+3)'s in s[92:95], V(t+2)'s in s[96:99] —
// so that moving one a tile forward is a handful of SALU instructions on its own words and the DMA reads the quad where it is: no
// copies into an aligned tuple, no second set of registers (the kernel sits at the SGPR limit; an SGPR spill costs a scratch access
// whose wait count the hand-counted DMA waits do not know).  base += tile bytes (with carry); rows left behind the new base -= 64;
// bound = clamp(rows left, 0, 64) x row bytes: past the sequence's end nothing is fetched.  Counting ROWS keeps every quantity far
// from 32 bits whatever the row stride (a layer's view of a megacache tensor has rows of 64 KiB and spans > 4 GiB at 128 k tokens).
// The caller places the call inside an MFMA gap.
__device__ __forceinline__ void k_rsrc_advance(u32x4& r, int& rows_left, unsigned tile_bytes, unsigned row_bytes) {
    asm volatile("s_add_u32 s92, s92, %3\n\ts_addc_u32 s93, s93, 0\n\ts_sub_i32 %1, %1, 64\n\ts_min_i32 s94, %1, 64\n\ts_max_i32 s94, s94, 0\n\ts_mul_i32 s94, s94, %4"
                 : "={s[92:95]}"(r), "+s"(rows_left) : "0"(r), "s"(tile_bytes), "s"(row_bytes) : "scc");
}
__device__ __forceinline__ void v_rsrc_advance(u32x4& r, int& rows_left, unsigned tile_bytes, unsigned row_bytes) {
    asm volatile("s_add_u32 s96, s96, %3\n\ts_addc_u32 s97, s97, 0\n\ts_sub_i32 %1, %1, 64\n\ts_min_i32 s98, %1, 64\n\ts_max_i32 s98, s98, 0\n\ts_mul_i32 s98, s98, %4"
                 : "={s[96:99]}"(r), "+s"(rows_left) : "0"(r), "s"(tile_bytes), "s"(row_bytes) : "scc");
}
__device__ __forceinline__ u32x4 tile_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;     // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane(bytes);                             // num_records (bytes): the bound
    r[3] = 0x00020000u;                                                       // raw buffer, 32-bit data format
    return r;
}

// ---- matrix instructions with the register FILE of every operand fixed by the constraint (cdna guide §5.7) ----
// S^T accumulators live in architectural VGPRs (the softmax VALU reads them), O^T accumulators and the Q^T fragments in the
// accumulator half of the 512-entry file; hipcc's own allocation of a 512-register kernel shuttles all of them through
// v_accvgpr_read/write (measured: 2 700 copies and 324 spills in the builtin version of this kernel).
// hipcc does not pad hazards around inline asm: callers keep MFMA results away from VALU readers by program order (an 8-pass
// MFMA result is readable >= 12 states later) and use the _NOP forms when an A/B/C operand was just written by the VALU.
template <typename T> struct Mfma;
#define VATTN_MFMA_STRUCT(TYPE, MNEM)                                                                                                    \
    template <> struct Mfma<TYPE> {                                                                                                \
        using V8 = typename Tr<TYPE>::v8;                                                                                          \
        /* S(vgpr) = A(vgpr) x B(agpr) + 0: the first MFMA of a chain */                                                           \
        static __device__ __forceinline__ void qk_first(f32x16& d, V8 a, V8 b) {                                                   \
            asm volatile(MNEM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));                                                       \
        }                                                                                                                          \
        static __device__ __forceinline__ void qk_acc(f32x16& d, V8 a, V8 b) {                                                     \
            asm volatile(MNEM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));                                                       \
        }                                                                                                                          \
        /* the same with the A fragment in the accumulator half (fragments carried from one tile step to the next are read from LDS  \
           straight into accumulator registers: no v_accvgpr_read at the seam) */                                                   \
        static __device__ __forceinline__ void qk_first_a(f32x16& d, V8 a, V8 b) {                                                 \
            asm volatile(MNEM " %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b));                                                       \
        }                                                                                                                          \
        static __device__ __forceinline__ void qk_acc_a(f32x16& d, V8 a, V8 b) {                                                   \
            asm volatile(MNEM " %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));                                                       \
        }                                                                                                                          \
        /* O(agpr) += A(vgpr) x B(vgpr) */                                                                                         \
        static __device__ __forceinline__ void pv(f32x16& o, V8 a, V8 b) {                                                         \
            asm volatile(MNEM " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));                                                       \
        }                                                                                                                          \
    };
VATTN_MFMA_STRUCT(_Float16, "v_mfma_f32_32x32x16_f16")
VATTN_MFMA_STRUCT(__bf16, "v_mfma_f32_32x32x16_bf16")
#undef VATTN_MFMA_STRUCT
// one scalar f32 add that the SLP vectoriser cannot pack into v_pk_add_f32 (packed f32 VALU beside MFMAs costs more than two
// plain adds, MI355X_MICROARCH "price of one filler")
__device__ __forceinline__ float add1(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// exp2 IN PLACE: the S' register becomes the P register (the builtin form lets the allocator give P fresh registers, and a
// second copy of the tile's 64 scores does not fit the architectural half of the register file)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

}  // namespace vattn_k
