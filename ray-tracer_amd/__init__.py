"""ctypes access to librtx.so: the C ABI of include/rtx.h (GPU render loop) and the C view of
the host-side scene API of include/rt_scene.h.

There is no CPU fallback anywhere in this package: loading fails loudly when the library is
missing, and every render call fails loudly when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

from . import scene_api
from .scene_api import (AMBIENT, CHECKER3D, CONE, CUBE, CYLINDER, DIFFUSE, F32, F64, F64_MAX, GRADIENT, PLANE,
                        REFLECTIVE, REFRACTIVE_INDEX, RING, SHININESS, SOLID, SPECULAR, SPHERE, STRIPE, TEST,
                        TRANSPARENCY, Panic, SceneAPI)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_LIB selects an alternative build of the library (kernel tuning experiments)
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(_HERE, "lib", "librtx.so")

if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: build it with ./build.sh (python -c 'import __graft_entry__ as g; g.build()')"
                      % LIB_PATH)


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime: a second copy opens no device ("No HIP GPUs are
    available").  PyTorch ships its own libamdhip64.so and asks for it by a name that does not
    match the SONAME librtx.so links against, so loading librtx.so first and importing torch later
    would bring in two.  If a PyTorch install is present and not yet imported, its copy is loaded
    here, globally, before librtx.so: same SONAME, so librtx.so binds to it, and a later
    `import torch` finds it already mapped."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


_share_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


class Stats(C.Structure):
    """rtx_stats"""
    _fields_ = [("color_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("rays_per_level", C.c_uint64 * 8),
                ("kernel_ms", C.c_double), ("primary_kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("n_launches", C.c_uint32), ("n_chunks", C.c_uint32), ("levels_ms", C.c_double)]

    def as_dict(self):
        return dict(color_rays=self.color_rays, shadow_rays=self.shadow_rays,
                    rays_per_level=list(self.rays_per_level), kernel_ms=self.kernel_ms,
                    primary_kernel_ms=self.primary_kernel_ms, total_ms=self.total_ms,
                    n_launches=self.n_launches, n_chunks=self.n_chunks, levels_ms=self.levels_ms)


class Probe(C.Structure):
    """rtx_probe"""
    _fields_ = [("t", C.c_double), ("point", C.c_double * 3), ("normalv", C.c_double * 3),
                ("over_point", C.c_double * 3), ("under_point", C.c_double * 3), ("reflectv", C.c_double * 3),
                ("n1", C.c_double), ("n2", C.c_double), ("lighting", C.c_double * 3), ("object", C.c_int32),
                ("inside", C.c_int32), ("shadowed", C.c_int32), ("pad", C.c_int32)]


PROBE_DTYPE = np.dtype([("t", "f8"), ("point", "f8", 3), ("normalv", "f8", 3), ("over_point", "f8", 3),
                        ("under_point", "f8", 3), ("reflectv", "f8", 3), ("n1", "f8"), ("n2", "f8"),
                        ("lighting", "f8", 3), ("object", "i4"), ("inside", "i4"), ("shadowed", "i4"),
                        ("pad", "i4")])
assert PROBE_DTYPE.itemsize == C.sizeof(Probe)


class SceneDesc(C.Structure):
    """rtx_scene_desc"""
    _fields_ = [("n_objects", C.c_uint32), ("shape_type", _u8p), ("inverse", _dp), ("inverse_transposed", _dp),
                ("ymin", _dp), ("ymax", _dp), ("closed", _u8p), ("color", _dp), ("ambient", _dp), ("diffuse", _dp),
                ("specular", _dp), ("shininess", _dp), ("reflective", _dp), ("transparency", _dp),
                ("refractive_index", _dp), ("casts_shadow", _u8p), ("pattern_type", _u8p), ("pattern_inverse", _dp),
                ("pattern_color1", _dp), ("pattern_color2", _dp), ("eq_class", _u32p), ("n_lights", C.c_uint32),
                ("light_position", _dp), ("light_intensity", _dp), ("chain_depth", _u8p), ("chain_first", _u32p),
                ("chain_inverse", _dp)]


class CameraDesc(C.Structure):
    """rtx_camera_desc"""
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("inverse_transform", C.c_double * 16),
                ("pixel_size", C.c_double), ("half_width", C.c_double), ("half_height", C.c_double)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint64, C.c_void_p)


def _bind(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


_vp = C.c_void_p
rtx_last_error = _bind("rtx_last_error", C.c_char_p, [])
rtx_context_create = _bind("rtx_context_create", C.c_int, [C.c_int, C.POINTER(_vp)])
rtx_context_destroy = _bind("rtx_context_destroy", None, [_vp])
rtx_context_synchronize = _bind("rtx_context_synchronize", C.c_int, [_vp])
rtx_context_set_debug_queue_divisor = _bind("rtx_context_set_debug_queue_divisor", C.c_int, [_vp, C.c_uint32])
rtx_context_set_workspace_limit = _bind("rtx_context_set_workspace_limit", C.c_int, [_vp, C.c_size_t])
rtx_context_set_measured_queues = _bind("rtx_context_set_measured_queues", C.c_int, [_vp, C.c_size_t])
rtx_context_queue_bytes = _bind("rtx_context_queue_bytes", C.c_int, [_vp, C.POINTER(C.c_size_t)])
rtx_context_set_cull_threshold = _bind("rtx_context_set_cull_threshold", C.c_int, [_vp, C.c_uint32])
rtx_context_set_lane_walk = _bind("rtx_context_set_lane_walk", C.c_int, [_vp, C.c_int])
rtx_scene_upload = _bind("rtx_scene_upload", C.c_int, [_vp, C.POINTER(SceneDesc), C.POINTER(_vp)])
rtx_scene_destroy = _bind("rtx_scene_destroy", None, [_vp])
rtx_scene_features = _bind("rtx_scene_features", C.c_int, [_vp, C.POINTER(C.c_uint32)])
rtx_render = _bind("rtx_render", C.c_int, [_vp, _vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                           _vp, C.c_size_t, _vp, _vp, C.POINTER(Stats)])
# pixel formats of rtx_render_8bit (include/rtx.h)
PIXELS_RGB8, PIXELS_RGBA8, PIXELS_PPM8 = 3, 4, 0x103
rtx_render_8bit = _bind("rtx_render_8bit", C.c_int, [_vp, _vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                                     C.c_uint32, _vp, C.c_size_t, _vp, _vp, C.POINTER(Stats)])
rtx_render_device = _bind("rtx_render_device", C.c_int, [_vp, _vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, C.c_uint32,
                                                         C.c_uint32, _vp, C.c_size_t, _vp, C.POINTER(Stats)])
rtx_part_rows = _bind("rtx_part_rows", C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32])
rtx_render_part_device = _bind("rtx_render_part_device", C.c_int, [_vp, _vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, C.c_uint32,
                                                                   C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_size_t, _vp,
                                                                   C.POINTER(Stats)])
rtx_render_device_noalloc = _bind("rtx_render_device_noalloc", C.c_int, [_vp, _vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int,
                                                                         C.c_uint32, C.c_uint32, _vp, C.c_size_t, _vp])
rtx_trace_rays = _bind("rtx_trace_rays", C.c_int, [_vp, _vp, _dp, C.c_size_t, C.c_uint32, C.c_int, _dp, C.POINTER(Stats)])
rtx_probe_rays = _bind("rtx_probe_rays", C.c_int, [_vp, _vp, _dp, C.c_size_t, _vp])
rtx_selfcheck_division = _bind("rtx_selfcheck_division", C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)])
rtx_selfcheck_pow = _bind("rtx_selfcheck_pow", C.c_int, [_vp, C.c_uint64, C.c_uint32, _dp, _dp])
rtx_generate_rays = _bind("rtx_generate_rays", C.c_int, [_vp, C.POINTER(CameraDesc), C.c_uint64, C.c_size_t, _dp])
rtx_quantize_u8_device = _bind("rtx_quantize_u8_device", C.c_int, [_vp, _vp, C.c_int, C.c_size_t, _vp, _vp])


class MultiStats(C.Structure):
    """rtx_multi_stats"""
    _fields_ = [("color_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("render_ms", C.c_double), ("gather_ms", C.c_double),
                ("total_ms", C.c_double), ("gather_bytes", C.c_uint64), ("n_devices", C.c_uint32), ("pad", C.c_uint32)]


MULTI_COUNT_RAYS = 1
rtx_multi_last_error = _bind("rtx_multi_last_error", C.c_char_p, [])
rtx_multi_create = _bind("rtx_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)])
rtx_multi_destroy = _bind("rtx_multi_destroy", None, [_vp])
rtx_multi_device_count = _bind("rtx_multi_device_count", C.c_int, [_vp])
rtx_multi_scene_upload = _bind("rtx_multi_scene_upload", C.c_int, [_vp, C.POINTER(SceneDesc), C.POINTER(_vp)])
rtx_multi_scene_destroy = _bind("rtx_multi_scene_destroy", None, [_vp])
rtx_multi_render = _bind("rtx_multi_render", C.c_int, [_vp, _vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, C.c_uint32, _vp, C.c_size_t,
                                                       C.c_int, C.c_uint32, C.POINTER(MultiStats)])
rtx_device_info = _bind("rtx_device_info", C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)])

# every symbol include/rtx.h declares
RTX_SYMBOLS = ["rtx_last_error", "rtx_context_create", "rtx_context_destroy", "rtx_context_synchronize",
               "rtx_context_set_debug_queue_divisor", "rtx_context_set_workspace_limit",
               "rtx_context_set_cull_threshold", "rtx_context_set_lane_walk", "rtx_context_set_measured_queues", "rtx_context_queue_bytes",
               "rtx_scene_upload", "rtx_scene_destroy", "rtx_scene_features", "rtx_render", "rtx_render_8bit", "rtx_render_device", "rtx_part_rows", "rtx_render_part_device", "rtx_render_device_noalloc", "rtx_trace_rays",
               "rtx_probe_rays", "rtx_selfcheck_division", "rtx_selfcheck_pow", "rtx_generate_rays", "rtx_quantize_u8_device", "rtx_device_info", "rtx_multi_last_error", "rtx_multi_create",
               "rtx_multi_destroy", "rtx_multi_device_count", "rtx_multi_scene_upload", "rtx_multi_scene_destroy", "rtx_multi_render"]


class RtxError(RuntimeError):
    pass


def _check(status):
    if status != 0:
        raise RtxError(rtx_last_error().decode("utf-8", "replace"))


# ---- host-side scene API (include/rt_scene.h) -------------------------------------------------
api = SceneAPI(lib, "rt_")

_rt_world_color_at_batch = _bind("rt_world_color_at_batch", C.c_int, [_vp, _dp, C.c_uint64, C.c_uint32, C.c_int, _dp,
                                                                        C.POINTER(C.c_uint64)])
_rt_world_probe = _bind("rt_world_probe", C.c_int, [_vp, _dp, C.c_uint64, _vp])
_rt_world_flat = _bind("rt_world_flat", C.c_int64, [_vp, C.c_int, _dp, C.c_int64])
_rt_camera_ray_for_pixel = _bind("rt_camera_ray_for_pixel", C.c_int, [_vp, C.c_uint64, C.c_uint64, _dp])
_rt_camera_render_rows = _bind("rt_camera_render_rows", C.c_int, [_vp, _vp, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64,
                                                                  _dp, C.POINTER(Stats)])
_rt_camera_parallel_render = _bind("rt_camera_parallel_render", C.c_int, [_vp, _vp, _dp, C.POINTER(C.c_uint64),
                                                                          C.POINTER(Stats)])

FLAT_INVERSE, FLAT_INVERSE_T, FLAT_PATTERN_INVERSE, FLAT_COLOR, FLAT_MATERIAL, FLAT_YRANGE, FLAT_TAGS, FLAT_LIGHTS, \
    FLAT_PATTERN_COLORS, FLAT_CHAIN, FLAT_CHAIN_INVERSE = range(11)


def _rays(rays):
    a = np.ascontiguousarray(np.asarray(rays, dtype=np.float64).reshape(-1, 6))
    return a


def color_at(world, rays, remaining=5, precision=F64, counters=None):
    """World::color_at for a batch of rays, on the GPU (src/scenes/world.rs:136-144)."""
    r = _rays(rays)
    out = np.empty((r.shape[0], 3))
    cnt = (C.c_uint64 * 2)(0, 0)
    api.check(_rt_world_color_at_batch(world.h, r.ctypes.data_as(_dp), r.shape[0], remaining, precision,
                                       out.ctypes.data_as(_dp), cnt))
    if counters is not None:
        counters[0] += cnt[0]
        counters[1] += cnt[1]
    return out


def probe(world, rays):
    """Intermediate results of one color_at node per ray (f64 path)."""
    r = _rays(rays)
    out = np.zeros(r.shape[0], dtype=PROBE_DTYPE)
    api.check(_rt_world_probe(world.h, r.ctypes.data_as(_dp), r.shape[0], out.ctypes.data_as(_vp)))
    return out


def flat(world, field):
    n = _rt_world_flat(world.h, field, None, 0)
    out = np.empty(max(n, 0))
    _rt_world_flat(world.h, field, out.ctypes.data_as(_dp), n)
    return out


def ray_for_pixel(camera, x, y):
    out = np.empty(6)
    api.check(_rt_camera_ray_for_pixel(camera.h, x, y, out.ctypes.data_as(_dp)))
    return out


def render_rows(camera, world, depth=5, precision=F64, row_begin=0, row_end=None, stats=None):
    """Camera::render restricted to rows [row_begin, row_end): array (rows, hsize, 3)."""
    g = camera.get()
    if row_end is None:
        row_end = g["vsize"]
    out = np.empty((row_end - row_begin, g["hsize"], 3))
    st = Stats()
    api.check(_rt_camera_render_rows(camera.h, world.h, depth, precision, row_begin, row_end, out.ctypes.data_as(_dp),
                                     C.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return out


def parallel_render(camera, world, stats=None):
    """Camera::parallel_render (depth 5) with the pixel tracker; returns (image, pixels_tracked)."""
    g = camera.get()
    out = np.empty((g["vsize"], g["hsize"], 3))
    tracker = C.c_uint64(0)
    st = Stats()
    api.check(_rt_camera_parallel_render(camera.h, world.h, out.ctypes.data_as(_dp), C.byref(tracker), C.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return out, tracker.value


# ---- direct use of the C ABI (what a Rust shim would do) ------------------------------------------
class Context:
    def __init__(self, device_id=-1):
        h = _vp()
        _check(rtx_context_create(device_id, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            rtx_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(rtx_context_synchronize(self.h))

    def set_debug_queue_divisor(self, divisor):
        _check(rtx_context_set_debug_queue_divisor(self.h, divisor))

    def set_workspace_limit(self, nbytes):
        _check(rtx_context_set_workspace_limit(self.h, nbytes))

    def set_measured_queues(self, worst_case_bytes_above):
        _check(rtx_context_set_measured_queues(self.h, worst_case_bytes_above))

    def queue_bytes(self):
        n = C.c_size_t()
        _check(rtx_context_queue_bytes(self.h, C.byref(n)))
        return n.value

    def set_cull_threshold(self, min_bounded_objects):
        _check(rtx_context_set_cull_threshold(self.h, min_bounded_objects))

    def set_lane_walk(self, enabled):
        _check(rtx_context_set_lane_walk(self.h, 1 if enabled else 0))

    def info(self):
        name = C.create_string_buffer(256)
        cu, mhz = C.c_int(), C.c_int()
        _check(rtx_device_info(self.h, name, 256, C.byref(cu), C.byref(mhz)))
        return dict(name=name.value.decode(), compute_units=cu.value, clock_mhz=mhz.value)


def flatten(world):
    """The structure-of-arrays World of rtx_scene_desc, as numpy arrays taken from the host
    mirror's own flattener (rt::World::flatten)."""
    n = len(world)
    tags = flat(world, FLAT_TAGS).reshape(n, 5)
    mat = flat(world, FLAT_MATERIAL).reshape(n, 7)
    yr = flat(world, FLAT_YRANGE).reshape(n, 2)
    lights = flat(world, FLAT_LIGHTS).reshape(-1, 6)
    pc = flat(world, FLAT_PATTERN_COLORS).reshape(n, 6)
    d = dict(
        shape_type=tags[:, 0].astype(np.uint8), closed=tags[:, 1].astype(np.uint8),
        casts_shadow=tags[:, 2].astype(np.uint8), pattern_type=tags[:, 3].astype(np.uint8),
        eq_class=tags[:, 4].astype(np.uint32),
        inverse=flat(world, FLAT_INVERSE), inverse_transposed=flat(world, FLAT_INVERSE_T),
        pattern_inverse=flat(world, FLAT_PATTERN_INVERSE), color=flat(world, FLAT_COLOR),
        ymin=np.ascontiguousarray(yr[:, 0]), ymax=np.ascontiguousarray(yr[:, 1]),
        ambient=np.ascontiguousarray(mat[:, 0]), diffuse=np.ascontiguousarray(mat[:, 1]),
        specular=np.ascontiguousarray(mat[:, 2]), shininess=np.ascontiguousarray(mat[:, 3]),
        reflective=np.ascontiguousarray(mat[:, 4]), transparency=np.ascontiguousarray(mat[:, 5]),
        refractive_index=np.ascontiguousarray(mat[:, 6]),
        pattern_color1=np.ascontiguousarray(pc[:, :3]).reshape(-1),
        pattern_color2=np.ascontiguousarray(pc[:, 3:]).reshape(-1),
        light_position=np.ascontiguousarray(lights[:, :3]).reshape(-1),
        light_intensity=np.ascontiguousarray(lights[:, 3:]).reshape(-1),
    )
    chain = flat(world, FLAT_CHAIN).reshape(n, 2)
    if chain[:, 0].any():
        d.update(chain_depth=chain[:, 0].astype(np.uint8), chain_first=chain[:, 1].astype(np.uint32),
                 chain_inverse=flat(world, FLAT_CHAIN_INVERSE))
    return d


def scene_desc(arrays):
    """(rtx_scene_desc, the arrays it points into) from a dict of numpy arrays named after its fields."""
    a = {k: (None if v is None else np.ascontiguousarray(v)) for k, v in arrays.items()}
    n = int(a["shape_type"].size)
    if a.get("pattern_color1") is None:
        a["pattern_color1"] = np.tile(np.array([1.0, 1.0, 1.0]), n)
    if a.get("pattern_color2") is None:
        a["pattern_color2"] = np.zeros(3 * n)
    d = SceneDesc()
    d.n_objects = n
    for name, ctype in SceneDesc._fields_:
        if name in ("n_objects", "n_lights"):
            continue
        v = a.get(name)
        if v is None:
            setattr(d, name, None)
            continue
        want = {_dp: np.float64, _u8p: np.uint8, _u32p: np.uint32}[ctype]
        v = np.ascontiguousarray(v, dtype=want)
        a[name] = v
        setattr(d, name, v.ctypes.data_as(ctype))
    d.n_lights = int(a["light_position"].size // 3)
    return d, a


class MultiError(RuntimeError):
    pass


class Multi:
    """Several GPUs rendering one frame (rtx_multi)."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = _vp()
        self._check(rtx_multi_create(ids, len(device_ids), C.byref(h)))
        self.h = h
        self.n = len(device_ids)

    @staticmethod
    def _check(status):
        if status != 0:
            raise MultiError(rtx_multi_last_error().decode("utf-8", "replace"))

    def upload(self, arrays):
        d, keep = scene_desc(arrays)
        h = _vp()
        self._check(rtx_multi_scene_upload(self.h, C.byref(d), C.byref(h)))
        return h

    def render(self, scene_h, cam_desc, depth, precision, fmt, out_ptr, nbytes, on_device=False, flags=0, stats=None):
        st = MultiStats() if stats is not None else None
        self._check(rtx_multi_render(self.h, scene_h, C.byref(cam_desc), depth, precision, fmt, _vp(out_ptr), nbytes,
                                     1 if on_device else 0, flags, C.byref(st) if st is not None else None))
        if stats is not None:
            stats.update({k: getattr(st, k) for k, _ in MultiStats._fields_})

    def close(self):
        if self.h:
            rtx_multi_destroy(self.h)
            self.h = None


class Scene:
    """An uploaded scene (rtx_scene).  `arrays` is a dict of numpy arrays named after the
    rtx_scene_desc fields."""

    def __init__(self, ctx, arrays):
        self.ctx = ctx
        a = {k: (None if v is None else np.ascontiguousarray(v)) for k, v in arrays.items()}
        n = int(a["shape_type"].size)
        if a.get("pattern_color1") is None:
            a["pattern_color1"] = np.tile(np.array([1.0, 1.0, 1.0]), n)
        if a.get("pattern_color2") is None:
            a["pattern_color2"] = np.zeros(3 * n)
        self._keep = a
        d = SceneDesc()
        d.n_objects = n
        for name, ctype in SceneDesc._fields_:
            if name in ("n_objects", "n_lights"):
                continue
            v = a.get(name)
            if v is None:
                setattr(d, name, None)
                continue
            want = {_dp: np.float64, _u8p: np.uint8, _u32p: np.uint32}[ctype]
            v = np.ascontiguousarray(v, dtype=want)
            a[name] = v
            setattr(d, name, v.ctypes.data_as(ctype))
        d.n_lights = int(a["light_position"].size // 3)
        h = _vp()
        _check(rtx_scene_upload(ctx.h, C.byref(d), C.byref(h)))
        self.h = h

    def features(self):
        """Feature bits of the kernel specialisation this scene selected (rtx_scene_features)."""
        bits = C.c_uint32()
        _check(rtx_scene_features(self.h, C.byref(bits)))
        v = bits.value
        return dict(reflective=bool(v & 1), transparent=bool(v & 2), all_cast_shadows=bool(v & 4), patterns=bool(v & 8),
                    bounding_sphere_tests=bool(v & 16), groups=bool(v & 32), lane_walk=bool(v & 64))

    def close(self):
        if self.h:
            rtx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_desc(camera):
    g = camera.get()
    d = CameraDesc()
    d.hsize, d.vsize = g["hsize"], g["vsize"]
    for i, v in enumerate(g["inverse"].reshape(-1)):
        d.inverse_transform[i] = v
    d.pixel_size, d.half_width, d.half_height = g["pixel_size"], g["half_width"], g["half_height"]
    return d


def render_part_device(ctx, scene, cam_desc, depth, precision, band_rows, n_parts, part, fmt, d_ptr, nbytes, stream=None, stats=None):
    """rtx_render_part_device: the bands of one part of the frame, values (fmt 0) or 8-bit pixels, in device memory."""
    st = Stats() if stats is not None else None
    _check(rtx_render_part_device(ctx.h, scene.h, C.byref(cam_desc), depth, precision, band_rows, n_parts, part, fmt, _vp(d_ptr),
                                  nbytes, _vp(stream) if stream else None, C.byref(st) if st is not None else None))
    if stats is not None:
        stats.update(st.as_dict())


def render_device(ctx, scene, cam_desc, depth, precision, row_begin, row_end, d_ptr, nbytes, stream=None, stats=None):
    """rtx_render_device: output stays in device memory at address d_ptr."""
    st = Stats() if stats is not None else None
    _check(rtx_render_device(ctx.h, scene.h, C.byref(cam_desc), depth, precision, row_begin, row_end, _vp(d_ptr), nbytes,
                             _vp(stream) if stream else None, C.byref(st) if st is not None else None))
    if stats is not None:
        stats.update(st.as_dict())
