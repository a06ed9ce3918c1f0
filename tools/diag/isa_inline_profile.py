"""Static ISA attribution through the inline stack (container, no GPU).
  hipcc ... -gline-tables-only -c --cuda-device-only -o kg.co ptk_kernels.hip; clang-offload-bundler --unbundle ... -> kg.elf
  python3 tools/diag/isa_inline_profile.py kg.elf <kernel substring> [depth]
Every instruction of the kernel is charged to the chain of the outermost `depth` inlined frames (default 2: the function called directly
from the kernel body, or the kernel's own line when nothing is inlined there).  VALU / SALU / LDS / VMEM counts per chain, static."""
import collections, json, re, subprocess, sys
L = "/opt/rocm/lib/llvm/bin/"
elf, pat = sys.argv[1], sys.argv[2]
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
syms = subprocess.run([L + "llvm-objdump", "-t", elf], capture_output=True, text=True).stdout.split("\n")
sym = [s.split()[-1] for s in syms if pat in s and " F " in s and not s.split()[-1].endswith(".kd")]
assert sym, "kernel not found"
dis = subprocess.run([L + "llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym[0], elf], capture_output=True, text=True).stdout
ins = []
for l in dis.split("\n"):
    m = re.match(r"\s+([a-z_0-9]+)\s.*//\s*([0-9A-F]+):", l) or re.match(r"\s+([a-z_0-9]+)\s*//\s*([0-9A-F]+):", l)
    if m: ins.append((int(m.group(2), 16), m.group(1)))
out = subprocess.run([L + "llvm-symbolizer", "--obj=" + elf, "--inlines", "--output-style=JSON"], input="\n".join(hex(a) for a, _ in ins), capture_output=True, text=True).stdout
recs = [json.loads(x) for x in out.split("\n") if x.strip()]
assert len(recs) == len(ins), (len(recs), len(ins))
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"
cnt = collections.defaultdict(collections.Counter)
for (a, op), r in zip(ins, recs):
    fr = r.get("Symbol", [])          # innermost first
    chain = []
    for f in reversed(fr[:-1] if len(fr) > 1 else fr):      # outermost inlined callee first (skip the kernel frame itself)
        chain.append(re.sub(r"<.*", "", f.get("FunctionName", "?")))
    klin = fr[-1].get("Line", 0) if fr else 0
    key = " > ".join(chain[:depth - 1]) if len(fr) > 1 else "(kernel body)"
    key = f"L{klin:4d} {key}"
    cnt[key][kind(op)] += 1
tot = collections.Counter()
for c in cnt.values(): tot.update(c)
print("total", dict(tot))
for key, c in sorted(cnt.items(), key=lambda kv: -kv[1]["valu"])[:70]:
    print(f"{c['valu']:6d} valu {c['salu']:5d} salu {c['lds']:4d} lds {c['vmem']:4d} vmem  {key}")
