#!/usr/bin/env python3
"""tools/isa_summary.py FILE.s KERNEL_SUBSTRING  -- static instruction mix of one gfx950 kernel from hipcc -save-temps output:
totals by class, basic blocks with their instruction counts (a loop body = the blocks between a label and the branch back to it),
register / LDS / occupancy figures from the kernel descriptor.  Used for profiles/rNN_score_kd_isa.txt."""
import collections, re, sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(l.split(":")[0] and l[l.index(":"):]) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]

def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"):
        if "_f64" in op: return "valu_f64"
        if op.startswith(("v_pk_",)): return "valu_pk_f32"
        if op.startswith(("v_cmp", "v_cndmask")): return "valu_cmp_sel"
        if "_f32" in op and not op.startswith("v_cvt"): return "valu_f32"
        if op.startswith("v_cvt"): return "valu_cvt"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mov", "v_accvgpr")): return "valu_mov"
        return "valu_int"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_getpc")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memrealtime") or op.startswith("s_memtime"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"

tot = collections.Counter()
blocks, cur, name = [], collections.Counter(), "entry"
edges = []
for l in body[1:]:
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1)
        continue
    op = t.split()[0]
    k = klass(op)
    tot[k] += 1; cur[k] += 1
    if k == "branch":
        m = re.search(r"(\.LBB\d+_\d+)", t)
        if m: edges.append((name, m.group(1)))
blocks.append((name, cur))
print("kernel: %s" % body[0].split(":")[0][:90])
n = sum(tot.values())
print("static instructions: %d   " % n + "  ".join("%s %d" % (k, v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
valu = sum(v for k, v in tot.items() if k.startswith("valu"))
print("VALU %d (f64 %d, f32 %d + packed %d, int %d, cmp/select %d, cvt %d, mov %d)  SALU %d  SMEM %d  VMEM %d  LDS %d  waits %d  branches %d" % (
    valu, tot["valu_f64"], tot["valu_f32"], tot["valu_pk_f32"], tot["valu_int"], tot["valu_cmp_sel"], tot["valu_cvt"], tot["valu_mov"],
    tot["salu"], tot["smem"], tot["vmem"], tot["lds"], tot["wait"], tot["branch"]))
order = {b[0]: i for i, b in enumerate(blocks)}
loops = sorted({(order[dst], order[srcb]) for srcb, dst in edges if dst in order and order[dst] <= order[srcb]})
print("loops (label .. back edge; static instructions inside):")
for a, b in loops:
    c = collections.Counter()
    for _, bc in blocks[a:b + 1]:
        c.update(bc)
    v = sum(x for k, x in c.items() if k.startswith("valu"))
    print("  %-12s .. %-12s %5d instr  VALU %4d (f64 %3d)  SALU %4d  SMEM %3d  VMEM %3d  LDS %3d  wait %3d  branch %3d" % (
        blocks[a][0], blocks[b][0], sum(c.values()), v, c["valu_f64"], c["salu"], c["smem"], c["vmem"], c["lds"], c["wait"], c["branch"]))
for l in lines[end:end + 400]:
    if any(s in l for s in (".sgpr_count", ".vgpr_count", ".agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", ".wavefront_size", "; Occupancy", "; NumVgprs", "; NumSgprs", "; ScratchSize", "; LDSByteSize")):
        print(l.strip())
    if l.strip().startswith(".end_amdhsa_kernel") or (l.startswith("_Z") and ":" in l):
        break
