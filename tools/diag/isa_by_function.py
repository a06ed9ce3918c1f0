"""Static ISA attribution (container, no GPU): compile ptk_kernels.hip with -gline-tables-only -S and count, for one kernel, the VALU /
SALU / LDS / VMEM instructions per innermost source function (by .loc line -> enclosing function of ptk_kernels.hip / ptk_device.hpp).
usage: python3 tools/diag/isa_by_function.py <asm.s> <kernel-name-substring> [--lines]"""
import re, sys, collections, os
asm, pat = sys.argv[1], sys.argv[2]
by_line = "--lines" in sys.argv
files = {}
src_funcs = {}

def funcs_of(path):
    if path in src_funcs: return src_funcs[path]
    out = []
    try: lines = open(path).read().split("\n")
    except OSError: src_funcs[path] = out; return out
    rx = re.compile(r"^(?:template.*>\s*)?(?:static\s+)?(?:__device__|__global__|__host__|inline).*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(")
    for i, l in enumerate(lines, 1):
        m = rx.match(l)
        if m and not l.strip().endswith(";"): out.append((i, m.group(1)))
    src_funcs[path] = out
    return out

def func_at(path, line):
    best = "?"
    for i, name in funcs_of(path):
        if i <= line: best = name
        else: break
    return best

cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
inside = False; cur = ("?", 0)
for l in open(asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if not m.group(3).startswith("/") else m.group(3); continue
    if not inside:
        if re.match(r"^_Z\S*:", l) and pat in l.split(":")[0]: inside = True
        continue
    if l.startswith(".Lfunc_end"): break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
    m = re.match(r"\s+([a-z_0-9]+)\s", l)
    if not m: continue
    op = m.group(1)
    if op.startswith("v_"): k = "valu"
    elif op.startswith("s_"): k = "salu"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "vmem"
    else: continue
    base = os.path.basename(cur[0])
    key = f"{base}:{cur[1]}" if by_line else f"{base}:{func_at(cur[0], cur[1])}"
    cnt[key] += 1; kinds[key][k] += 1
tot = collections.Counter()
for k in kinds.values(): tot.update(k)
print("total", dict(tot))
for key, n in cnt.most_common(60):
    print(f"{n:6d}  {key:60s} {dict(kinds[key])}")
