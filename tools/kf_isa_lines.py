#!/usr/bin/env python3
"""tools/kf_isa_lines.py ASM [KERNEL_SRC] -- static count of the vector / scalar / LDS / memory instructions of one kernel's assembly
(compiled with -gline-tables-only) per OUTERMOST k_frame.hip line of the kernel body and per innermost function: where the
instructions of k_frame sit, without weights (tools/isa_classes.py prices them)."""
import re, sys, collections
asm = open(sys.argv[1]).read().splitlines()
src = open(sys.argv[2] if len(sys.argv) > 2 else "mlvfs_amd/csrc/k_frame.hip").read().splitlines()
# function of a source line: nearest preceding line that looks like a function / lambda header
heads = []
for i, l in enumerate(src, 1):
    m = re.match(r"^(?:template.*>\s*)?(?:__device__|__global__|static|MLV_NET_FN).*?\b([A-Za-z_0-9]+)\s*\(", l)
    if m: heads.append((i, m.group(1)))
    m = re.match(r"^\s+auto ([a-z_0-9]+) = \[", l)
    if m: heads.append((i, "λ" + m.group(1)))
def fn_of(line):
    best = "?"
    for i, n in heads:
        if i <= line: best = n
        else: break
    return best
loc = re.compile(r"k_frame\.hip:(\d+):\d+")
cur_chain = []
by_outer = collections.Counter(); by_fn = collections.Counter(); kinds = collections.Counter()
by_outer_fn = collections.Counter()
for ln in asm:
    if "\t.loc\t" in ln:
        cur_chain = [int(x) for x in loc.findall(ln.split(";", 1)[1])] if ";" in ln else []
        continue
    m = re.match(r"^\t([a-z][a-z0-9_]+)\b", ln)
    if not m or ln.startswith("\t."): continue
    op = m.group(1)
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_")) else "other"
    kinds[kind] += 1
    if kind != "valu": continue
    outer = cur_chain[-1] if cur_chain else 0
    inner = cur_chain[0] if cur_chain else 0
    by_outer[outer] += 1
    by_fn[fn_of(inner)] += 1
    by_outer_fn[(outer, fn_of(inner))] += 1
print("instructions:", dict(kinds))
print("\nVALU by innermost function:")
for f, n in by_fn.most_common(40): print(f"  {n:6d}  {f}")
print("\nVALU by outermost line (kernel body) and innermost function:")
for (o, f), n in sorted(by_outer_fn.items()):
    if n >= 8: print(f"  line {o:5d} {n:6d}  {f:28s} | {src[o-1].strip()[:90] if 0 < o <= len(src) else ''}")
