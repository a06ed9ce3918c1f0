"""GPU diagnostic (needs a libptk.so built with EXTRA_HIPFLAGS=-DPTK_PHASE_TIMING): where the waves of k_intersect and
k_shade_binned spend their lifetime, in shader-clock cycles per wave and phase (s_memtime stamps, s_waitcnt(0) at every stamp)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
scn = os.path.join(ROOT, "scenes", sys.argv[1] if len(sys.argv) > 1 else "cornell.txt")
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = ptk.Scene(scn)
r = ptk.Renderer(s, 0, in_flight=lanes, graph=lanes > 1, pipeline=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
for it in range(1, 4): r.iterate(it)
r.sync()
out = np.zeros(16, np.uint64)
ptk.lib.ptk_debug_phases.argtypes = [C.c_void_p]
ptk.lib.ptk_debug_phases(out.ctypes.data)
for it in range(4, 24): r.iterate(it)
r.sync()
ptk.lib.ptk_debug_phases(out.ctypes.data)
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
names = [["counts+windows", "stage geoms", "ray loads", "candidates+pairs+exact", "hit record+stores", "aggregates"],
         ["staging", "flags+slot+sort", "loads+shading", "compaction+stores", "segments"]]
if pipe == 0:      # the fused kernel's shading launches (blocks with survivors) use slot 0 of the phase buffer (PTK_PHASE markers in k_shade_trace)
    names[0] = ["counts + geoms -> LDS", "materials, tables, word, sort", "record loads + shading", "survivor ranks + stores", "closest hit (candidates/exact)", "hit record + map + stores", "tail: counts, done counters"]
for k, kn in enumerate(("k_shade_trace" if pipe == 0 else "k_intersect", "k_shade_binned")):
    c = int(out[k * 8 + 7]); tot = float(out[k * 8:k * 8 + 7].sum())
    print(f"{kn}: {c} waves, {tot / max(c, 1):.0f} cycles per wave")
    for j, nm in enumerate(names[k]):
        v = float(out[k * 8 + j])
        print(f"   {nm:26s} {v / max(c, 1):9.0f} cycles  {100 * v / max(tot, 1):5.1f} %")
r.close(); s.close()
