"""Static instruction counts between the trace points of a kernel (analysis build: tools/kernel_regs.sh <unit>.hip -DCADNIP_MARKS leaves
the assembly in /tmp/isa).  One wave alone pays ~7 cycles per instruction, so the counts are the latency model of the team kernel.

usage: python tools/phase_instr.py /tmp/isa/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s <kernel name substring>
Counts are in program (layout) order from one mark to the next: code of both sides of a branch is counted."""
import re, sys, collections


def main():
    path, kern = sys.argv[1], sys.argv[2]
    inside = False
    cur = "entry"
    counts = collections.OrderedDict()
    kinds = collections.defaultdict(lambda: collections.Counter())
    for line in open(path):
        if not inside:
            if re.match(r"^\S*%s\S*:" % re.escape(kern), line):
                inside = True
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            break
        m = re.match(r"; @@MARK (\d+)", t)
        if m:
            cur = "after mark %s" % m.group(1)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        counts[cur] = counts.get(cur, 0) + 1
        k = ("branch" if op.startswith("s_cbranch") or op == "s_branch" else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else
             "smem" if op.startswith("s_load") else "spill" if op in ("v_readlane_b32", "v_writelane_b32") else "wait" if op in ("s_waitcnt", "s_nop", "s_barrier") else
             "salu" if op.startswith("s_") else "valu")
        kinds[cur][k] += 1
    print("%-16s %6s   %s" % ("region", "instr", "valu salu lds vmem smem spill wait branch"))
    for k, v in counts.items():
        c = kinds[k]
        print("%-16s %6d   %4d %4d %3d %4d %4d %5d %4d %6d" % (k, v, c["valu"], c["salu"], c["lds"], c["vmem"], c["smem"], c["spill"], c["wait"], c["branch"]))


if __name__ == "__main__":
    main()
