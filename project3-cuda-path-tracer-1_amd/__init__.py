"""project3-cuda-path-tracer-1_amd — MI355X-native per-bounce path-tracing inner loop.

The product is the C++/HIP library ``csrc/libptk.so`` (hand-written gfx950 kernels + the C++
host scene layer + the C ABI declared in ``include/ptk.h``).  This module is only the thin
ctypes binding that the tests and ``bench.py`` use to drive that C ABI from Python; torch is
used purely as plumbing (device buffers via ``tensor.data_ptr()``, streams, torch.distributed).

There is NO fallback: if ``libptk.so`` is missing the import raises, and every compute entry
point returns a non-zero code (raised as ``PtkError``) when no GPU / scene is available.
Nothing here imports, links or executes anything under ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# PTK_FLAVOUR=fma selects the opt-in "reference-device numerics" build (libptk_fma.so: -ffp-contract=fast, like nvcc's -fmad=true);
# PTK_FLAVOUR=hooks the test-hooks build (libptk_hooks.so: the product sources + include/ptk_test_hooks.h, used by child processes of the tests);
# anything else is the default bit-exact library.  One flavour per process (the library holds the render context).
FLAVOUR = os.environ.get("PTK_FLAVOUR", "") or "exact"       # "fma", or the name of a diagnostic build libptk_<name>.so (tools/diag)
if FLAVOUR != "exact" and not os.path.exists(os.path.join(CSRC, f"libptk_{FLAVOUR}.so")):
    raise ImportError(f"PTK_FLAVOUR={FLAVOUR}: {os.path.join(CSRC, f'libptk_{FLAVOUR}.so')} does not exist (make -C csrc / tools/diag/build_variant.sh)")
LIB_PATH = os.path.join(CSRC, "libptk.so" if FLAVOUR == "exact" else f"libptk_{FLAVOUR}.so")
REPO_ROOT = os.path.dirname(_HERE)
SCENES = os.path.join(REPO_ROOT, "scenes")

MODE_BSDF, MODE_DIRECTLI, MODE_MIS = 0, 1, 2
GEOM_SPHERE, GEOM_CUBE, GEOM_OBJ = 0, 1, 2
MAT_LAMBERTIAN, MAT_METALLIC, MAT_DIELECTRIC, MAT_MICROFACET, MAT_LIGHT = 0, 1, 2, 3, 4
E_BADARG, E_NOSCENE, E_PARSE, E_NODEVICE, E_DEVICE = 1001, 1002, 1003, 1004, 1005

# every symbol include/ptk.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = [
    "ptk_scene_load", "ptk_scene_free", "ptk_scene_error", "ptk_scene_counts", "ptk_scene_get_camera",
    "ptk_scene_get_geoms", "ptk_scene_get_materials", "ptk_scene_get_triangles", "ptk_scene_get_bvh",
    "ptk_scene_get_lights", "ptk_scene_bvh_device_bytes", "ptk_scene_wide_bvh_counts", "ptk_scene_get_wide_bvh", "ptk_scene_wide_bvh_counts_w", "ptk_scene_get_wide_bvh_w", "ptk_scene_get_texture_size", "ptk_scene_get_texture", "ptk_scene_upload", "ptk_generate_rays", "ptk_intersect", "ptk_compact",
    "ptk_shade", "ptk_tonemap", "ptk_pathtrace_init", "ptk_pathtrace_free", "ptk_pathtrace",
    "ptk_set_sample_mode", "ptk_get_sample_mode", "ptk_get_image", "ptk_get_image_device",
    "ptk_copy_image_to_device", "ptk_get_live_counts", "ptk_get_ray_total", "ptk_set_tile_partition",
    "ptk_set_pipeline", "ptk_stream_sync", "ptk_timer_start", "ptk_timer_stop",
    "ptk_profile_begin", "ptk_profile_end", "ptk_enable_traversal_stats", "ptk_get_traversal_stats",
    "ptk_partition_count", "ptk_partition_pixel", "ptk_set_iterations_in_flight", "ptk_set_ordered_accumulation", "ptk_set_tiles_in_flight", "ptk_set_graph_replay", "ptk_set_samples_per_launch", "ptk_set_light_fix",
    "ptk_set_iteration_sharding", "ptk_get_full_width_bounces", "ptk_get_open_scene_hint",
]
# include/ptk_test_hooks.h: exported by libptk_hooks.so (PTK_FLAVOUR=hooks) only, never by the product library
HOOK_SYMBOLS = ["ptk_debug_slot_shift", "ptk_debug_clear_image", "ptk_debug_exact_math"]


class PtkError(RuntimeError):
    def __init__(self, what: str, code: int):
        super().__init__(f"{what} failed with code {code}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile libptk.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "libptk.so", "libptk_fma.so", "libptk_hooks.so", "ptk_headless"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libptk.so failed")
    return LIB_PATH


class PtkCamera(C.Structure):
    _fields_ = [("res", C.c_int * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3), ("view", C.c_float * 3),
                ("up", C.c_float * 3), ("right", C.c_float * 3), ("fov", C.c_float * 2), ("pixelLength", C.c_float * 2)]


class PtkPaths(C.Structure):
    _fields_ = [("o_pix", C.c_void_p), ("d_bnc", C.c_void_p), ("col_pdf", C.c_void_p)]


class PtkHits(C.Structure):
    _fields_ = [("t_n", C.c_void_p), ("p_mat", C.c_void_p), ("uv_ids", C.c_void_p)]


# numpy views of the interchange PODs (include/ptk.h), dword-for-dword
GEOM_DT = np.dtype([("type", "<i4"), ("materialid", "<i4"), ("transform", "<f4", 16), ("inverseTransform", "<f4", 16), ("invTranspose", "<f4", 16)])
MATERIAL_DT = np.dtype([("type", "<i4"), ("albedo", "<f4", 3), ("albedo_s", "<f4", 3), ("rough_s", "<f4"), ("metal_s", "<f4"), ("ior", "<f4"),
                        ("albedoMapID", "<i4"), ("metallicMapID", "<i4"), ("roughnessMapID", "<i4"), ("normalMapID", "<i4")])
TRIANGLE_DT = np.dtype([("v", "<f4", 9), ("n", "<f4", 9), ("tex", "<f4", 6), ("tangent", "<f4", 3), ("bitangent", "<f4", 3), ("geomIdx", "<i4")])
WIDE_LEAF_DT = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("v0", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("tri", "<i4")])
BVHNODE_DT = np.dtype([("pMin", "<f4", 3), ("pMax", "<f4", 3), ("start", "<i4"), ("end", "<i4"), ("miss", "<i4"), ("hit", "<i4")])
LIGHT_DT = np.dtype([("geomID", "<i4"), ("triangleID", "<i4"), ("type", "<i4")])
CAMERA_DT = np.dtype([("res", "<i4", 2), ("position", "<f4", 3), ("lookAt", "<f4", 3), ("view", "<f4", 3), ("up", "<f4", 3),
                      ("right", "<f4", 3), ("fov", "<f4", 2), ("pixelLength", "<f4", 2)])


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.  Two HIP runtimes in one process cannot share stream handles,
    so make torch's copy the one libptk.so binds to: load it (by file) before libptk.so, whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the HIP path)")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    lib.ptk_scene_load.argtypes = [C.c_char_p, ci, ci, ci, C.POINTER(vp)]
    lib.ptk_scene_free.argtypes = [vp]; lib.ptk_scene_free.restype = None
    lib.ptk_scene_error.argtypes = [vp]; lib.ptk_scene_error.restype = C.c_char_p
    for name in ("counts", "get_camera", "get_geoms", "get_materials", "get_triangles", "get_bvh", "get_lights"):
        getattr(lib, "ptk_scene_" + name).argtypes = [vp, vp]
    lib.ptk_scene_upload.argtypes = [vp]
    lib.ptk_scene_wide_bvh_counts.argtypes = [vp, vp]
    lib.ptk_scene_bvh_device_bytes.argtypes = [vp, vp]
    lib.ptk_scene_get_wide_bvh.argtypes = [vp, vp, vp]
    lib.ptk_scene_wide_bvh_counts_w.argtypes = [vp, C.c_int, vp]
    lib.ptk_scene_get_wide_bvh_w.argtypes = [vp, C.c_int, vp, vp]
    lib.ptk_scene_get_texture_size.argtypes = [vp, ci, vp, vp]
    lib.ptk_scene_get_texture.argtypes = [vp, ci, vp]
    lib.ptk_generate_rays.argtypes = [C.POINTER(PtkCamera), ci, ci, PtkPaths, vp]
    lib.ptk_intersect.argtypes = [vp, ci, vp, PtkPaths, PtkHits, vp, vp, vp, vp]
    lib.ptk_compact.argtypes = [ci, vp, vp, PtkPaths, PtkHits, PtkPaths, PtkHits, vp, vp]
    lib.ptk_shade.argtypes = [vp, ci, ci, ci, ci, vp, PtkHits, PtkPaths, vp, vp, vp]
    lib.ptk_tonemap.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.ptk_pathtrace_init.argtypes = [vp]
    lib.ptk_pathtrace_free.argtypes = []
    lib.ptk_pathtrace.argtypes = [vp, ci, ci]
    lib.ptk_set_sample_mode.argtypes = [ci]
    lib.ptk_get_image.argtypes = [vp]
    lib.ptk_get_image_device.argtypes = [C.POINTER(vp)]
    lib.ptk_copy_image_to_device.argtypes = [vp]
    lib.ptk_get_live_counts.argtypes = [vp, C.POINTER(ci)]
    lib.ptk_get_ray_total.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.ptk_set_tile_partition.argtypes = [ci, ci, ci]
    lib.ptk_set_pipeline.argtypes = [ci]
    lib.ptk_set_iterations_in_flight.argtypes = [ci]
    lib.ptk_set_tiles_in_flight.argtypes = [ci]
    lib.ptk_set_ordered_accumulation.argtypes = [ci]
    lib.ptk_set_graph_replay.argtypes = [ci]
    lib.ptk_set_samples_per_launch.argtypes = [ci]
    lib.ptk_set_light_fix.argtypes = [ci]
    lib.ptk_set_iteration_sharding.argtypes = [ci, ci]
    lib.ptk_get_full_width_bounces.argtypes = [C.POINTER(ci)]
    lib.ptk_get_open_scene_hint.argtypes = [C.POINTER(ci)]
    if FLAVOUR == "hooks":
        lib.ptk_debug_slot_shift.argtypes = [ci, ci]
        lib.ptk_debug_exact_math.argtypes = [C.c_ulonglong, C.c_uint, vp, vp]
    lib.ptk_timer_stop.argtypes = [C.POINTER(C.c_float)]
    lib.ptk_profile_end.argtypes = [vp, vp]
    lib.ptk_enable_traversal_stats.argtypes = [ci]
    lib.ptk_get_traversal_stats.argtypes = [vp]
    return lib


lib = _load()


def _chk(code: int, what: str) -> None:
    if code != 0:
        raise PtkError(what, code)


class Scene:
    """Host scene (class Scene of the reference: parser + OBJ + BVH + lights + viewer camera fix-ups)."""

    def __init__(self, path: str, res=None, depth: int = -1):
        h = C.c_void_p()
        w, hh = (res if res else (-1, -1))
        rc = lib.ptk_scene_load(os.fsencode(path), int(w), int(hh), int(depth), C.byref(h))
        self.handle = h
        if rc != 0:
            msg = lib.ptk_scene_error(h).decode() if h else "no scene"
            lib.ptk_scene_free(h)
            self.handle = None
            raise PtkError(f"ptk_scene_load({path}): {msg}", rc)
        c = np.zeros(11, np.int32)
        _chk(lib.ptk_scene_counts(h, c.ctypes.data), "ptk_scene_counts")
        (self.n_geoms, self.n_materials, self.n_triangles, self.bvh_size, self.n_lights,
         self.trace_depth, self.iterations, self.width, self.height, self.n_textures, self.env_map_id) = (int(x) for x in c)
        self.uploaded = False

    def _table(self, getter, dtype, n):
        out = np.zeros(max(n, 0), dtype)
        if n > 0:
            _chk(getter(self.handle, out.ctypes.data), getter.__name__)
        return out

    def camera(self): return self._table(lib.ptk_scene_get_camera, CAMERA_DT, 1)[0]
    def geoms(self): return self._table(lib.ptk_scene_get_geoms, GEOM_DT, self.n_geoms)
    def materials(self): return self._table(lib.ptk_scene_get_materials, MATERIAL_DT, self.n_materials)
    def triangles(self): return self._table(lib.ptk_scene_get_triangles, TRIANGLE_DT, self.n_triangles)
    def bvh(self): return self._table(lib.ptk_scene_get_bvh, BVHNODE_DT, 6 * self.bvh_size)
    def lights(self): return self._table(lib.ptk_scene_get_lights, LIGHT_DT, self.n_lights)

    def wide_bvh(self, width: int = 8):
        """-> (nodes uint32 [n, 36 | 68], leaves (WIDE_LEAF_DT) [m], depth, stack bound) of the 8-wide (or 16-wide) collapsed tree; n = 0 if there is none"""
        c = np.zeros(4, np.int32)
        _chk(lib.ptk_scene_wide_bvh_counts_w(self.handle, width, c.ctypes.data), "ptk_scene_wide_bvh_counts_w")
        nodes = np.zeros((int(c[0]), 4 + 4 * width), np.uint32); leaves = np.zeros(int(c[1]), WIDE_LEAF_DT)
        if c[0] > 0:
            _chk(lib.ptk_scene_get_wide_bvh_w(self.handle, width, nodes.ctypes.data, leaves.ctypes.data), "ptk_scene_get_wide_bvh_w")
        return nodes, leaves, int(c[2]), int(c[3])

    def bvh_device_bytes(self):
        """-> (threaded nodes, wide nodes, wide leaves, rank table) bytes on the device (after upload())"""
        out = np.zeros(4, np.uint64)
        _chk(lib.ptk_scene_bvh_device_bytes(self.handle, out.ctypes.data), "ptk_scene_bvh_device_bytes")
        return tuple(int(x) for x in out)

    def texture(self, i: int) -> np.ndarray:
        """float32 [h, w, 3], row 0 = bottom row (as the reference loads it)"""
        w, h = C.c_int(0), C.c_int(0)
        _chk(lib.ptk_scene_get_texture_size(self.handle, i, C.byref(w), C.byref(h)), "ptk_scene_get_texture_size")
        out = np.zeros((h.value, w.value, 3), np.float32)
        _chk(lib.ptk_scene_get_texture(self.handle, i, out.ctypes.data), "ptk_scene_get_texture")
        return out

    def camera_struct(self) -> PtkCamera:
        cam = PtkCamera()
        _chk(lib.ptk_scene_get_camera(self.handle, C.addressof(cam)), "ptk_scene_get_camera")
        return cam

    def upload(self):
        _chk(lib.ptk_scene_upload(self.handle), "ptk_scene_upload")
        self.uploaded = True
        return self

    def close(self):
        if self.handle:
            lib.ptk_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_pixels(W: int, H: int, n_parts: int, part: int, tile_rows: int) -> np.ndarray:
    """Pixel indices (x + y*W) owned by `part`, in the order its paths are generated (host-only helper)."""
    n = lib.ptk_partition_count(W, H, n_parts, part, tile_rows)
    if n < 0:
        raise PtkError("ptk_partition_count", E_BADARG)
    return np.array([lib.ptk_partition_pixel(W, n_parts, part, tile_rows, j) for j in range(0, n, W)], np.int64)[:, None].repeat(W, 1) \
        + np.arange(W, dtype=np.int64)[None, :] if n else np.zeros((0, W), np.int64)


# ---------------------------------------------------------------------------------------------
# kernel-level wrappers (torch tensors on the GPU are only device buffers here)
# ---------------------------------------------------------------------------------------------
def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def new_paths(n, device="cuda"):
    import torch
    return [torch.empty((max(n, 1), 4), dtype=torch.float32, device=device) for _ in range(3)]


def paths_struct(bufs) -> PtkPaths:
    return PtkPaths(*(C.c_void_p(b.data_ptr()) for b in bufs)) if bufs is not None else PtkPaths(None, None, None)


def hits_struct(bufs) -> PtkHits:
    return PtkHits(*(C.c_void_p(b.data_ptr()) for b in bufs)) if bufs is not None else PtkHits(None, None, None)


def generate_rays(scene: Scene, it: int, paths, trace_depth=None):
    cam = scene.camera_struct()
    _chk(lib.ptk_generate_rays(C.byref(cam), it, scene.trace_depth if trace_depth is None else trace_depth,
                               paths_struct(paths), _stream()), "ptk_generate_rays")


def intersect(scene: Scene, n: int, paths, hits, valid, d_n=None, stats=None, img=None):
    _chk(lib.ptk_intersect(scene.handle, n, _ptr(d_n), paths_struct(paths), hits_struct(hits), _ptr(valid), _ptr(stats), _ptr(img), _stream()), "ptk_intersect")


def compact(n: int, valid, paths_in, hits_in, paths_out, hits_out, d_count_out, d_n=None):
    _chk(lib.ptk_compact(n, _ptr(d_n), _ptr(valid), paths_struct(paths_in), hits_struct(hits_in), paths_struct(paths_out),
                         hits_struct(hits_out), _ptr(d_count_out), _stream()), "ptk_compact")


def shade(scene: Scene, mode: int, it: int, depth: int, n: int, hits, paths, valid, img, d_n=None):
    _chk(lib.ptk_shade(scene.handle, mode, it, depth, n, _ptr(d_n), hits_struct(hits), paths_struct(paths), _ptr(valid), _ptr(img), _stream()), "ptk_shade")


def tonemap(img, w: int, h: int, it: int, pbo):
    _chk(lib.ptk_tonemap(_ptr(img), w, h, it, _ptr(pbo), _stream()), "ptk_tonemap")


# ---------------------------------------------------------------------------------------------
# render API (pathtraceInit / pathtrace / pathtraceFree through the C ABI)
# ---------------------------------------------------------------------------------------------
def set_light_fix(on: bool) -> None:
    """Opt-in fixed cube lights (ptk_set_light_fix; not reference behaviour, default off)."""
    _chk(lib.ptk_set_light_fix(1 if on else 0), "ptk_set_light_fix")


class Renderer:
    """File-static render context of the library (one at a time, like the reference)."""

    def __init__(self, scene: Scene, mode: int = MODE_MIS, partition=(1, 0, 1), pipeline: int = 0, in_flight: int = 1,
                 graph: bool = False, spp: int = 1, shard=(1, 0), tiles: int = 1, ordered: int = -1):
        """shard = (n, g): iterate(step) renders iteration (step-1)*n + g + 1 of the whole frame (ptk_set_iteration_sharding);
        tiles = K: K bands of rows in flight, bit-identical to the default (ptk_set_tiles_in_flight);
        ordered: accumulation with several planes (ptk_set_ordered_accumulation; -1 / 1 = ordered, the default; 0 = planes that accumulate)"""
        self.scene = scene
        lib.ptk_pathtrace_free()
        _chk(lib.ptk_set_iteration_sharding(*shard), "ptk_set_iteration_sharding")
        _chk(lib.ptk_set_samples_per_launch(spp), "ptk_set_samples_per_launch")
        _chk(lib.ptk_set_graph_replay(1 if graph else 0), "ptk_set_graph_replay")
        _chk(lib.ptk_set_iterations_in_flight(in_flight), "ptk_set_iterations_in_flight")
        _chk(lib.ptk_set_tiles_in_flight(tiles), "ptk_set_tiles_in_flight")
        _chk(lib.ptk_set_ordered_accumulation(ordered), "ptk_set_ordered_accumulation")
        _chk(lib.ptk_set_pipeline(pipeline), "ptk_set_pipeline")
        _chk(lib.ptk_set_tile_partition(*partition), "ptk_set_tile_partition")
        _chk(lib.ptk_set_sample_mode(mode), "ptk_set_sample_mode")
        if not scene.uploaded:
            scene.upload()
        _chk(lib.ptk_pathtrace_init(scene.handle), "ptk_pathtrace_init")
        self.P = scene.width * scene.height

    def iterate(self, it: int, pbo=None):
        _chk(lib.ptk_pathtrace(_ptr(pbo), 0, it), "ptk_pathtrace")

    def sync(self):
        _chk(lib.ptk_stream_sync(), "ptk_stream_sync")

    def image(self) -> np.ndarray:
        out = np.zeros((self.P, 3), np.float32)
        _chk(lib.ptk_get_image(out.ctypes.data), "ptk_get_image")
        return out

    def image_to(self, tensor):
        """radiance sum -> a caller-owned device tensor.  The render streams are non-blocking, so nothing orders work the caller queued
        on ITS stream (e.g. the zero-fill of a fresh torch.zeros) against the copy: wait for the caller's stream first."""
        import torch
        torch.cuda.current_stream(tensor.device).synchronize()
        _chk(lib.ptk_copy_image_to_device(_ptr(tensor)), "ptk_copy_image_to_device")

    def live_counts(self):
        """per-bounce live-path counts of the LAST iterate() call (all samples of a batched launch together)"""
        c = np.zeros(self.scene.trace_depth + 1, np.int32)
        d = C.c_int(0)
        _chk(lib.ptk_get_live_counts(c.ctypes.data, C.byref(d)), "ptk_get_live_counts")
        return c, d.value

    def full_width_bounces(self) -> int:
        d = C.c_int(0)
        _chk(lib.ptk_get_full_width_bounces(C.byref(d)), "ptk_get_full_width_bounces")
        return d.value

    def open_scene_hint(self) -> int:
        d = C.c_int(0)
        _chk(lib.ptk_get_open_scene_hint(C.byref(d)), "ptk_get_open_scene_hint")
        return d.value

    def ray_total(self) -> int:
        r = C.c_ulonglong(0)
        _chk(lib.ptk_get_ray_total(C.byref(r)), "ptk_get_ray_total")
        return int(r.value)

    def timer_start(self):
        _chk(lib.ptk_timer_start(), "ptk_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        _chk(lib.ptk_timer_stop(C.byref(ms)), "ptk_timer_stop")
        return float(ms.value)

    def profile_begin(self):
        _chk(lib.ptk_profile_begin(), "ptk_profile_begin")

    def profile_end(self):
        """-> {class: (total_ms, launches)} for generate / intersect / shade / compact (k_segments, or k_scatter in pipeline 4)"""
        ms = np.zeros(4, np.float32); n = np.zeros(4, np.int32)
        _chk(lib.ptk_profile_end(ms.ctypes.data, n.ctypes.data), "ptk_profile_end")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(("generate", "intersect", "shade", "compact"))}

    def traversal_stats(self, enable=None):
        if enable is not None:
            _chk(lib.ptk_enable_traversal_stats(1 if enable else 0), "ptk_enable_traversal_stats")
            return None
        out = np.zeros(2, np.uint64)
        _chk(lib.ptk_get_traversal_stats(out.ctypes.data), "ptk_get_traversal_stats")
        return int(out[0]), int(out[1])

    def close(self):
        lib.ptk_pathtrace_free()
        lib.ptk_set_tile_partition(1, 0, 1)
        lib.ptk_set_iterations_in_flight(1)
        lib.ptk_set_tiles_in_flight(1)
        lib.ptk_set_ordered_accumulation(-1)
        lib.ptk_set_graph_replay(0)
        lib.ptk_set_samples_per_launch(1)
        lib.ptk_set_iteration_sharding(1, 0)
