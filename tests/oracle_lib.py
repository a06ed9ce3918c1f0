"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY — imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

GEOM_DT = np.dtype([("type", "<i4"), ("materialid", "<i4"), ("transform", "<f4", 16), ("inverseTransform", "<f4", 16), ("invTranspose", "<f4", 16)])
MATERIAL_DT = np.dtype([("type", "<i4"), ("albedo", "<f4", 3), ("albedo_s", "<f4", 3), ("rough_s", "<f4"), ("metal_s", "<f4"), ("ior", "<f4"),
                        ("albedoMapID", "<i4"), ("metallicMapID", "<i4"), ("roughnessMapID", "<i4"), ("normalMapID", "<i4")])
TRIANGLE_DT = np.dtype([("v", "<f4", 9), ("n", "<f4", 9), ("tex", "<f4", 6), ("tangent", "<f4", 3), ("bitangent", "<f4", 3), ("geomIdx", "<i4")])
BVHNODE_DT = np.dtype([("pMin", "<f4", 3), ("pMax", "<f4", 3), ("start", "<i4"), ("end", "<i4"), ("miss", "<i4"), ("hit", "<i4")])
LIGHT_DT = np.dtype([("geomID", "<i4"), ("triangleID", "<i4"), ("type", "<i4")])
CAMERA_DT = np.dtype([("res", "<i4", 2), ("position", "<f4", 3), ("lookAt", "<f4", 3), ("view", "<f4", 3), ("up", "<f4", 3),
                      ("right", "<f4", 3), ("fov", "<f4", 2), ("pixelLength", "<f4", 2)])
# AoS path / hit records of the reference (sceneStructs.h:80-99), 12 dwords each
PATH_DT = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("color", "<f4", 3), ("pixelIndex", "<i4"), ("remainingBounces", "<i4"), ("prevPdf", "<f4")])
ISECT_DT = np.dtype([("t", "<f4"), ("n", "<f4", 3), ("p", "<f4", 3), ("uv", "<f4", 2), ("materialId", "<i4"), ("geomID", "<i4"), ("triangleID", "<i4")])

MODE_BSDF, MODE_DIRECTLI, MODE_MIS = 0, 1, 2
TRIG_LIBM, TRIG_PTK = 0, 1


def build():
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


if not os.path.exists(LIB_PATH):
    build()
L = C.CDLL(LIB_PATH)
vp, ci, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
L.orc_scene_load.restype = vp; L.orc_scene_load.argtypes = [C.c_char_p, ci, ci, ci]
L.orc_scene_free.argtypes = [vp]
L.orc_scene_error.restype = C.c_char_p; L.orc_scene_error.argtypes = [vp]
for _n in ("counts", "get_geoms", "get_materials", "get_triangles", "get_bvh", "get_lights", "get_camera"):
    getattr(L, "orc_scene_" + _n).argtypes = [vp, vp]
L.orc_scene_texture_path.restype = C.c_char_p; L.orc_scene_texture_path.argtypes = [vp, ci]
L.orc_scene_set_texture.argtypes = [vp, ci, ci, ci, vp]
L.orc_intersect_env.argtypes = [vp, ci, vp, vp, vp, vp]
L.orc_ptk_atan2_array.argtypes = [vp, vp, ci, vp]
L.orc_utilhash.restype = C.c_uint32; L.orc_utilhash.argtypes = [C.c_uint32]
L.orc_seed_hash.restype = C.c_uint32; L.orc_seed_hash.argtypes = [ci, ci, ci]
L.orc_rng_uniforms.argtypes = [ci, ci, ci, ci, vp]
L.orc_rng_uniforms_seed.argtypes = [C.c_uint32, ci, vp]
L.orc_ptk_sincos_array.argtypes = [vp, ci, vp, vp]
L.orc_box_test.restype = C.c_float; L.orc_box_test.argtypes = [vp, vp, vp, vp, vp]
L.orc_sphere_test.restype = C.c_float; L.orc_sphere_test.argtypes = [vp, vp, vp, vp, vp]
L.orc_box_test_array.argtypes = [vp, vp, vp, ci, vp]
L.orc_sphere_test_array.argtypes = [vp, vp, vp, ci, vp]
L.orc_aabb_test.argtypes = [vp, vp, vp, vp, vp]
L.orc_tri_test.argtypes = [vp, vp, vp, vp]
L.orc_make_geom.argtypes = [ci, ci, vp, vp, vp, vp]
L.orc_scatter_sample.argtypes = [vp, vp, vp, ci, ci, ci, vp]
L.orc_mat_bsdf_pdf.argtypes = [vp, vp, vp, vp, vp]
L.orc_light_sample.argtypes = [vp, vp, ci, ci, ci, ci, vp]
L.orc_light_pdf.restype = C.c_float; L.orc_light_pdf.argtypes = [vp, vp, vp, vp, ci, ci]
L.orc_generate_rays.argtypes = [vp, ci, vp]
L.orc_intersect.argtypes = [vp, ci, vp, vp, vp]
L.orc_shade.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp]
L.orc_compact.restype = ci; L.orc_compact.argtypes = [ci, vp, vp, vp]
L.orc_pathtrace.restype = C.c_longlong; L.orc_pathtrace.argtypes = [vp, ci, ci, vp, vp, vp]
L.orc_pathtrace_tiled.restype = C.c_longlong; L.orc_pathtrace_tiled.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp]
L.orc_pathtrace_batch.restype = C.c_longlong; L.orc_pathtrace_batch.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]
L.orc_stats.argtypes = [vp, ci]
L.orc_tonemap.argtypes = [vp, ci, ci, vp]
L.orc_set_trig_mode.argtypes = [ci]
L.orc_set_sample2d_order.argtypes = [ci]
L.orc_set_debug_slot_shift.argtypes = [ci]
L.orc_set_light_fix.argtypes = [ci]
L.orc_set_threads.argtypes = [ci]


def set_trig_mode(m):
    L.orc_set_trig_mode(m)


def set_threads(n):
    """threads of the per-path loops (results are identical for any count)"""
    L.orc_set_threads(int(n))


def set_light_fix(on):
    """opt-in fixed cube lights (not reference behaviour)"""
    L.orc_set_light_fix(1 if on else 0)


def f32(*v):
    return np.ascontiguousarray(np.array(v, np.float32).ravel())


class Scene:
    def __init__(self, path, res=None, depth=-1):
        w, h = res if res else (-1, -1)
        self.h = L.orc_scene_load(os.fsencode(path), int(w), int(h), int(depth))
        err = L.orc_scene_error(self.h)
        if err:
            msg = err.decode(); L.orc_scene_free(self.h); self.h = None
            raise RuntimeError(msg)
        c = np.zeros(11, np.int32); L.orc_scene_counts(self.h, c.ctypes.data)
        (self.n_geoms, self.n_materials, self.n_triangles, self.bvh_size, self.n_lights,
         self.trace_depth, self.iterations, self.width, self.height, self.n_textures, self.env_map_id) = (int(x) for x in c)
        self.P = self.width * self.height
        # the oracle has no image decoder: the harness decodes (PIL / image_ref.py) and hands the texels over
        import image_ref
        self.texture_data = []
        for i in range(self.n_textures):
            tex = image_ref.load_texture(L.orc_scene_texture_path(self.h, i).decode())
            self.texture_data.append(tex)
            L.orc_scene_set_texture(self.h, i, tex.shape[1], tex.shape[0], tex.ctypes.data)

    def _table(self, fn, dt, n):
        out = np.zeros(n, dt)
        if n:
            fn(self.h, out.ctypes.data)
        return out

    def geoms(self): return self._table(L.orc_scene_get_geoms, GEOM_DT, self.n_geoms)
    def materials(self): return self._table(L.orc_scene_get_materials, MATERIAL_DT, self.n_materials)
    def triangles(self): return self._table(L.orc_scene_get_triangles, TRIANGLE_DT, self.n_triangles)
    def bvh(self): return self._table(L.orc_scene_get_bvh, BVHNODE_DT, 6 * self.bvh_size)
    def lights(self): return self._table(L.orc_scene_get_lights, LIGHT_DT, self.n_lights)
    def camera(self): return self._table(L.orc_scene_get_camera, CAMERA_DT, 1)[0]

    def generate_rays(self, it):
        p = np.zeros(self.P, PATH_DT); L.orc_generate_rays(self.h, it, p.ctypes.data); return p

    def intersect(self, paths, img=None):
        """img (optional, float32 [P,3]): ENV-map radiance of the misses is added to it (pathtrace.cu:303-311)"""
        n = len(paths); paths = np.ascontiguousarray(paths)
        isects = np.zeros(n, ISECT_DT); valid = np.zeros(n, np.int32)
        L.orc_intersect_env(self.h, n, paths.ctypes.data, isects.ctypes.data, valid.ctypes.data, img.ctypes.data if img is not None else None)
        return isects, valid

    def shade(self, mode, it, depth, isects, paths, img):
        """shades in place (paths, img); returns valid"""
        n = len(paths); valid = np.zeros(n, np.int32)
        assert paths.flags.c_contiguous and isects.flags.c_contiguous and img.flags.c_contiguous
        L.orc_shade(self.h, mode, it, depth, n, isects.ctypes.data, paths.ctypes.data, valid.ctypes.data, img.ctypes.data)
        return valid

    def pathtrace(self, mode, it, img, partition=(1, 0, 1), timing=None):
        live = np.zeros(self.trace_depth + 1, np.int32)
        rays = L.orc_pathtrace_tiled(self.h, mode, it, partition[0], partition[1], partition[2], img.ctypes.data, live.ctypes.data,
                                     timing.ctypes.data if timing is not None else None)
        return int(rays), live

    def pathtrace_batch(self, mode, step, spp, planes, partition=(1, 0, 1)):
        """planes: float32 [spp, P, 3] contiguous; sample s of the launch accumulates into planes[s]"""
        assert planes.flags.c_contiguous and planes.shape == (spp, self.P, 3)
        live = np.zeros(self.trace_depth + 1, np.int32)
        rays = L.orc_pathtrace_batch(self.h, mode, step, spp, partition[0], partition[1], partition[2], planes.ctypes.data, live.ctypes.data, None)
        return int(rays), live

    def close(self):
        if self.h:
            L.orc_scene_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Accumulator:
    """The product's accumulation, restated for the checks (csrc/pathtrace.hip: merge_lane / resolve_image; include/ptk.h:
    ptk_set_ordered_accumulation).  ONE radiance plane (one iteration at a time, one sample per launch; tiles in flight): every add goes into the
    image itself, as in the reference (pathtrace.cu:398,486,556,572,600).  SEVERAL planes (L iterations in flight and / or S samples per launch):
      ordered (default) — a launch adds into planes of its own (one per sample; one per bounce when the context was initialised in MIS
                 mode with one sample per launch), zero when it starts, every (pixel, plane) receiving one add, and the planes are folded into the image in the reference's
                 order of adds when the launch is done: the image IS the plain running sum, whatever the mode.  (A context initialised in BSDF /
                 DIRECT that is switched to MIS later has one plane per sample: MIS iterations then add their own sum.)
      ordered=False — lane l's planes accumulate its calls l+1, l+1+L, ... and the image is ((plane_0 + plane_1) + plane_2) + ... over lanes x samples."""

    def __init__(self, P, lanes=1, spp=1, mode=0, ordered=True):
        self.lanes, self.spp = lanes, spp
        self.single = lanes * spp <= 1
        self.ordered = bool(ordered)
        self.per_bounce = mode == MODE_MIS and spp == 1      # planes per bounce: fixed at ptk_pathtrace_init by the sample mode of that moment
        self.image = np.zeros((P, 3), np.float32)
        self.planes = None if (self.single or self.ordered) else np.zeros((lanes, spp, P, 3), np.float32)

    def _direct(self, mode):
        return self.single or (self.ordered and (self.per_bounce or mode != MODE_MIS))

    def iteration(self, scene, mode, it, partition=(1, 0, 1), step=None):
        """one call of ptk_pathtrace(.., step) with one sample per launch; it = the iteration it renders (== step unless iterations are sharded)"""
        assert self.spp == 1
        step = it if step is None else step
        if self._direct(mode):
            return scene.pathtrace(mode, it, self.image, partition=partition)
        if self.ordered:
            tmp = np.zeros_like(self.image)
            r = scene.pathtrace(mode, it, tmp, partition=partition)
            self.image += tmp
            return r
        return scene.pathtrace(mode, it, self.planes[(step - 1) % self.lanes][0], partition=partition)

    def batch(self, scene, mode, step, partition=(1, 0, 1)):
        """one call of ptk_pathtrace(.., step) with spp samples per launch (planes per bounce are a one-sample-per-launch feature: with spp > 1
        every sample has one plane, folded into the image in sample order)"""
        if self.spp == 1 and self._direct(mode):
            return scene.pathtrace_batch(mode, step, 1, self.image.reshape(1, -1, 3), partition=partition)
        if self.ordered:
            tmp = np.zeros((self.spp,) + self.image.shape, np.float32)
            r = scene.pathtrace_batch(mode, step, self.spp, tmp, partition=partition)
            for s in range(self.spp):
                self.image += tmp[s]
            return r
        return scene.pathtrace_batch(mode, step, self.spp, self.planes[(step - 1) % self.lanes], partition=partition)

    @property
    def img(self):
        if self.planes is None:
            return self.image
        flat = self.planes.reshape(self.lanes * self.spp, -1, 3)
        out = flat[0].copy()
        for p in flat[1:]:
            out = out + p
        return out


def compact(valid, paths, isects=None):
    keep = valid.astype(bool)
    return paths[keep].copy(), (isects[keep].copy() if isects is not None else None)


def traversal_stats(reset=True):
    st = np.zeros(2, np.uint64); L.orc_stats(st.ctypes.data, 1 if reset else 0); return int(st[0]), int(st[1])


# ---- SoA <-> AoS conversion between the product's float4 streams and the reference-shaped records ----
def paths_to_soa(p):
    n = len(p)
    o = np.zeros((n, 4), np.float32); d = np.zeros((n, 4), np.float32); c = np.zeros((n, 4), np.float32)
    o[:, :3] = p["o"]; o[:, 3] = p["pixelIndex"].view(np.float32)
    d[:, :3] = p["d"]; d[:, 3] = p["remainingBounces"].view(np.float32)
    c[:, :3] = p["color"]; c[:, 3] = p["prevPdf"]
    return o, d, c


def paths_from_soa(o, d, c):
    n = len(o); p = np.zeros(n, PATH_DT)
    p["o"] = o[:, :3]; p["pixelIndex"] = np.ascontiguousarray(o[:, 3]).view(np.int32)
    p["d"] = d[:, :3]; p["remainingBounces"] = np.ascontiguousarray(d[:, 3]).view(np.int32)
    p["color"] = c[:, :3]; p["prevPdf"] = c[:, 3]
    return p


def isects_to_soa(h):
    n = len(h)
    a = np.zeros((n, 4), np.float32); b = np.zeros((n, 4), np.float32); c = np.zeros((n, 4), np.float32)
    a[:, 0] = h["t"]; a[:, 1:] = h["n"]
    b[:, :3] = h["p"]; b[:, 3] = h["materialId"].view(np.float32)
    c[:, :2] = h["uv"]; c[:, 2] = h["geomID"].view(np.float32); c[:, 3] = h["triangleID"].view(np.float32)
    return a, b, c


def isects_from_soa(a, b, c):
    n = len(a); h = np.zeros(n, ISECT_DT)
    h["t"] = a[:, 0]; h["n"] = a[:, 1:]
    h["p"] = b[:, :3]; h["materialId"] = np.ascontiguousarray(b[:, 3]).view(np.int32)
    h["uv"] = c[:, :2]; h["geomID"] = np.ascontiguousarray(c[:, 2]).view(np.int32); h["triangleID"] = np.ascontiguousarray(c[:, 3]).view(np.int32)
    return h
