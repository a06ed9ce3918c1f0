/* ptk_test_hooks.h — TEST-ONLY entry points.  They are compiled into libptk_hooks.so only (the same sources with -DPTK_TEST_HOOKS,
 * csrc/Makefile); the product library libptk.so does not export them (tests/test_abi.py).  The tests that need them run in a child
 * process with PTK_FLAVOUR=hooks (tests/hooks_child.py). */
#ifndef PTK_TEST_HOOKS_H
#define PTK_TEST_HOOKS_H
#include "ptk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Offsets added to the compacted slot in the shading RNG key (pathtrace.cu:363,458,532): k0 at depth 0, k1 at depth >= 1.  tests/test_gallery.py
 * renders the reference's gallery/{Direct,BSDF,MIS}20.png with the slot offsets its GPU build ran with (fixture tables `slot_shift`,
 * `slot_shift_deep`).  ptk_debug_clear_image: reset of the radiance buffer, so that single iterations can be read back. */
int ptk_debug_slot_shift(int k0, int k1);
int ptk_debug_clear_image(void);
/* The kernels' exact fast paths for IEEE divide / sqrt / 1/sqrt and their packed 3-vector arithmetic (csrc/ptk_device.hpp) against the plain
 * `/`, sqrtf and scalar forms, bit for bit, on n pseudo-random operand sets on the device; bad8 (host) receives the mismatch counts of the
 * eight checks, samples128 (host, optional) the first failing operand sets ([0] = their number, then 8 words each: check mask, six operand
 * bit patterns, operand class).  tests/test_gpu_exact_math.py. */
int ptk_debug_exact_math(unsigned long long n, unsigned seed, unsigned long long* bad8, unsigned* samples128);

#ifdef __cplusplus
}
#endif
#endif
