"""GPU diagnostic (needs a library built with -DPTK_PHASE_TIMING, e.g. tools/diag/build_variant.sh phase -DPTK_PHASE_TIMING, PTK_FLAVOUR=phase): where the waves of k_tail spend
the launch, s_memtime ticks per launch and phase (stamps with s_waitcnt(0): upper bounds), wave 0 — the wave that holds the paths of a bounce of a few dozen.
usage: tail_phases.py [scene mode lanes W H depth]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
a = sys.argv[1:] or ["mis_test.txt", "2", "4", "1024", "1024", "16"]
s = ptk.Scene(os.path.join(ROOT, "scenes", a[0]), res=(int(a[3]), int(a[4])), depth=int(a[5]))
lanes = int(a[2])
r = ptk.Renderer(s, int(a[1]), in_flight=lanes, graph=lanes > 1)
for it in range(1, 65): r.iterate(it)
r.sync()
out = np.zeros(16, np.uint64)
ptk.lib.ptk_debug_phases.argtypes = [C.c_void_p]
ptk.lib.ptk_debug_phases(out.ctypes.data)
r.timer_start()
for it in range(65, 65 + 256): r.iterate(it)
ms = r.timer_stop()
ptk.lib.ptk_debug_phases(out.ctypes.data)
names = ["staging (primitives, materials, hit-prefix table)", "gather + path / hit-record loads", "closest hit (in lane)", "block prefixes", "shading", "survivor stores, counts, end-of-bounce barrier"]
k = 1
waves = int(out[k * 8 + 7]); tot = float(out[k * 8:k * 8 + 6].sum()); launches = waves
print(f"{' '.join(a)}: {1e3 * ms / 256:.1f} us per iteration; k_tail: {launches:.0f} launches, {float(out[k * 8 + 6]) / max(waves, 1):.2f} bounces per launch, {tot / max(waves, 1):.0f} ticks per launch (wave 0 of the block)")
for j, nm in enumerate(names):
    v = float(out[k * 8 + j])
    print(f"   {nm:52s} {v / max(waves, 1):9.0f} ticks  {100 * v / max(tot, 1):5.1f} %")
r.close(); s.close()
