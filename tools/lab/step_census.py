#!/usr/bin/env python3
"""LAB: census of the steady-state tile step of a prefill64 kernel from the device assembly (hipcc -S --offload-device-only).
usage: step_census.py dev.s [kernel-substring]   -> instruction classes per tile step (two steps per loop iteration) and per MFMA gap."""
import re, sys, collections

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_accvgpr") : return "accmov"
    if op.startswith("v_mov"): return "vmov"
    if op.startswith("v_exp"): return "exp"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_"): return "valu"
    return "other"

def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "DF16"
    lines = open(path).read().splitlines()
    # kernel bodies: from "<name>:" to ".Lfunc_end"
    start = None
    for i, l in enumerate(lines):
        if ("prefill64" in l or "prefill32" in l) and want in l.split(":")[0] and re.match(r"^_Z\w+:", l):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # the hot loop: the label that is the target of a backward branch and contains the most MFMAs
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            n = sum(1 for x in body[a:b] if "v_mfma" in x)
            if best is None or n > best[2]:
                best = (a, b, n)
    a, b, n = best
    cnt = collections.Counter()
    ops = collections.Counter()
    # rare blocks: whatever a forward conditional branch inside the loop jumps over (mask of the diagonal tile, rescale, loop exit)
    skip = set()
    for i in range(a, b + 1):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", body[i])
        if m and m.group(1) in labels and i < labels[m.group(1)] <= b:
            skip.update(range(i + 1, labels[m.group(1)]))
    print("rare (branched-over) lines inside the loop: %d" % len(skip))
    for i, l in enumerate(body[a:b + 1], a):
        if i in skip:
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        cnt[classify(op)] += 1
        ops[op] += 1
    total = sum(cnt.values())
    print("loop lines %d-%d of the kernel, %d MFMAs, %d instructions = %.2f issues per MFMA gap (incl. the MFMA)" % (start + a + 1, start + b + 1, n, total, total / n))
    for k, v in cnt.most_common():
        print("  %-8s %5d  %.2f per MFMA" % (k, v, v / n))
    nonmfma_valu = sum(v for k, v in cnt.items() if k in ("valu", "exp", "cvt", "vmov", "accmov"))
    print("non-MFMA VALU per MFMA: %.2f" % (nonmfma_valu / n))
    if "-v" in sys.argv:
        for k, v in ops.most_common():
            print("    %-28s %d" % (k, v))

main()


def gaps(path, want):
    """per MFMA gap of the hot loop: instruction count and a first-order issue cost (4 cycles per instruction, v_exp 8, s_nop N: N+1 states)"""
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if ("prefill64" in l or "prefill32" in l) and want in l.split(":")[0] and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            n = sum(1 for x in body[a:b] if "v_mfma" in x)
            if best is None or n > best[2]:
                best = (a, b, n)
    a, b, n = best
    skip = set()
    for i in range(a, b + 1):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", body[i])
        if m and m.group(1) in labels and i < labels[m.group(1)] <= b:
            skip.update(range(i + 1, labels[m.group(1)]))
    out, cur = [], None
    for i, l in enumerate(body[a:b + 1], a):
        if i in skip:
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            if cur is not None:
                out.append(cur)
            cur = {"n": 0, "cost": 4, "ops": []}
            continue
        if cur is None:
            continue
        cur["n"] += 1
        cur["cost"] += 8 if op.startswith("v_exp") else 4
        cur["ops"].append(op.replace("_e32", "").replace("_e64", ""))
    out.append(cur)
    return out


if "--gaps" in sys.argv:
    g = gaps(sys.argv[1], sys.argv[2])
    tot = sum(x["cost"] for x in g)
    model = sum(max(32, x["cost"]) for x in g)
    print("gaps %d: issue cost sum %d cycles (%.1f per gap), sum of max(32, cost) = %d (%.1f per gap)" % (len(g), tot, tot / len(g), model, model / len(g)))
    for i, x in enumerate(g):
        c = collections.Counter(x["ops"])
        print("  gap %3d  n=%2d cost=%3d  %s" % (i, x["n"], x["cost"], " ".join("%s%s" % (k, "x%d" % v if v > 1 else "") for k, v in c.items())))
