"""LAB ONLY (round 4, prefill64 energy pass (ii)).  Drops the `s_nop 0` the compiler puts between two inline-asm statements of
prefill64_kernel when no real instruction separates a producer asm from a consumer asm that reads one of its registers.

Why they are there: LLVM's hazard recogniser cannot see inside an inline-asm statement; on gfx950 it assumes the producer might end in a
dst_sel / cvt-scale instruction (whose result needs one wait state before it is forwarded) and counts inline-asm statements as ZERO wait
states, so  [asm: v_fma v58] [asm: v_mfma ...] [asm: v_exp v58]  gets a nop in front of the v_exp although a 4-cycle-issue MFMA sits
between the two and neither is a dst_sel instruction.  This script removes exactly those: an `s_nop 0` whose previous statement is an asm
block consisting of one v_mfma and whose next statement is an asm block of plain f32 VALU (v_fma / v_exp / v_add / v_max3 / v_max).
usage: strip_asm_nops.py in.s out.s  (prints the count)."""
import re
import sys

PLAIN = re.compile(r"^\s*(v_fma_f32|v_exp_f32|v_add_f32|v_max3_f32|v_max_f32|v_mul_f32|v_sub_f32)\b")


def main(src, dst):
    lines = open(src).read().split("\n")
    out, removed, in_kernel = [], 0, False
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_ZN7vattn_k16prefill64_kernel\S*):", l)
        if m:
            in_kernel = True
        elif in_kernel and "s_endpgm" in l:
            in_kernel = False
        if in_kernel and l.strip() == "s_nop 0":
            # previous statement: ;;#ASMEND closing a one-instruction v_mfma block
            j = len(out) - 1
            prev_ok = j >= 2 and out[j].strip() == ";;#ASMEND" and out[j - 1].strip().startswith("v_mfma") and out[j - 2].strip() == ";;#ASMSTART"
            k = i + 1
            nxt_ok = k < len(lines) and lines[k].strip() == ";;#ASMSTART"
            if nxt_ok:
                k += 1
                n = 0
                while k < len(lines) and lines[k].strip() != ";;#ASMEND":
                    if not PLAIN.match(lines[k]):
                        nxt_ok = False
                    n += 1
                    k += 1
                nxt_ok = nxt_ok and n > 0
            if prev_ok and nxt_ok:
                removed += 1
                i += 1
                continue
        out.append(l)
        i += 1
    open(dst, "w").write("\n".join(out))
    print("removed %d s_nop 0" % removed)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
