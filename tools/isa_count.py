"""Instruction counts of the loops of a kernel, from the device assembly the compiler writes.

usage: isa_count.py ASSEMBLY.s KERNEL_SUBSTRING [MIN_VMEM]
  (assembly: hipcc <the Makefile's flags> -S --offload-device-only kfh_kernels.hip -o ASSEMBLY.s)
Every backward branch of the kernel closes a loop; for each loop whose body holds at least MIN_VMEM (default 4)
vector-memory instructions the script prints the body's size by instruction class.  Nested loops are printed on
their own (the outer count includes the inner one's instructions once).  Packed f32 instructions are listed apart: on
gfx950 they issue at the rate of two plain ones.
"""
import re
import sys


def classify(op):
    if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store")):
        return "vmem_store"
    if op.startswith(("buffer_atomic", "global_atomic", "flat_atomic")):
        return "vmem_atomic"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_transcendental"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    min_vmem = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(kernel), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    labels, insts = {}, []
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        insts.append(t)
    print("%s: %d instructions" % (kernel, len(insts)))
    for i, t in enumerate(insts):
        parts = t.split()
        if parts[0].startswith(("s_cbranch", "s_branch")) and parts[-1] in labels and labels[parts[-1]] <= i:
            body = insts[labels[parts[-1]]:i + 1]
            counts = {}
            for b in body:
                c = classify(b.split()[0])
                counts[c] = counts.get(c, 0) + 1
            vmem = counts.get("vmem_load", 0) + counts.get("vmem_store", 0)
            if vmem < min_vmem:
                continue
            wide = sum(1 for b in body if re.match(r"buffer_(load|store)_dwordx2", b))
            print("loop %s .. +%d: %d instructions, %s%s" % (
                parts[-1], i - labels[parts[-1]], len(body),
                ", ".join("%s %d" % kv for kv in sorted(counts.items())),
                " (%d of the memory instructions move 8 bytes per lane)" % wide if wide else ""))


if __name__ == "__main__":
    main()
