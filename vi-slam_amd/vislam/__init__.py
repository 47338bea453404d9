"""vislam -- thin ctypes binding over the C ABI of libvislam_hip.so (include/vislam_hip.h).

This is plumbing for tests and bench.py; the product is the shared library.  The classes here
mirror the reference's surface for the hot path (names, argument meaning, error behaviour):

  Context.orb_detect_compute   <-> CameraGPU::detectAndComputeGPUFeatures  (src/CameraGPU.cpp:71-117)
  Context.bf_knn2_hamming      <-> MatcherGPU::computeGPUMatches           (src/MatcherGPU.cpp:44-66)
  Context.good_matches         <-> Matcher::computeBestMatches             (src/Matcher.cpp:353-367)
  Context.essential_ransac / recover_pose <-> VISystem::EstimatePoseFeaturesRansac (src/VISystem.cpp:1679-1701)
  Context.f2f_ransac           <-> VISystem::F2FRansac                      (src/VISystem.cpp:612-769)

There is NO CPU fallback: if the HIP library is missing, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

# PyTorch (plumbing for device memory / streams / torch.distributed in tests and bench.py) bundles its
# own libamdhip64.so.7.  Two HIP runtimes in one process cannot both own the GPU, so when torch is
# installed it is imported FIRST: the loader then resolves this library's libamdhip64.so.7 dependency
# to the copy that is already mapped.  The library itself has no torch dependency.
try:  # pragma: no cover
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# VISLAM_HIP_LIB: another build of the same library (A/B measurements); there is no fallback either way
LIB_PATH = os.environ.get("VISLAM_HIP_LIB") or os.path.join(_HERE, "..", "lib", "libvislam_hip.so")


class VisError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: error {code} ({_strerror(code)}) {detail}")


class Params(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("scale_factor", C.c_float),
        ("edge_threshold", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
        ("ratio", C.c_float), ("n_cells", C.c_int32), ("w_size", C.c_int32), ("h_size", C.c_int32),
        ("sym_mode", C.c_int32),
        ("ransac_prob", C.c_double), ("ransac_threshold", C.c_double),
        ("ransac_max_iters", C.c_int32), ("ransac_adaptive", C.c_int32), ("ransac_seed", C.c_uint64),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("f2f_iters", C.c_int32), ("f2f_threshold", C.c_double),
        ("pose_input", C.c_int32), ("keypoint_capacity", C.c_int32),
    ]

    def copy(self):
        p = Params()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Params))
        return p


class Se3f(C.Structure):
    """Sophus::SE3f storage order: unit quaternion (x, y, z, w) + translation"""
    _fields_ = [("qx", C.c_float), ("qy", C.c_float), ("qz", C.c_float), ("qw", C.c_float),
                ("tx", C.c_float), ("ty", C.c_float), ("tz", C.c_float)]

    def as_array(self):
        return np.array([self.qx, self.qy, self.qz, self.qw, self.tx, self.ty, self.tz], np.float32)


class AlignParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("first_level", C.c_int32), ("last_level", C.c_int32), ("max_iterations", C.c_int32),
                ("epsilon", C.c_float), ("z_factor", C.c_float)]


class AlignResult(C.Structure):
    _fields_ = [("pose", Se3f), ("matrix", C.c_float * 16), ("error", C.c_float * 5), ("initial_error", C.c_float),
                ("iterations", C.c_int32 * 5), ("n_residuals", C.c_int32 * 5)]


class PoseResult(C.Structure):
    _fields_ = [("E", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3), ("n_inliers", C.c_int32), ("n_pose_good", C.c_int32),
                ("iters_run", C.c_int32), ("n_points", C.c_int32), ("n_models", C.c_int32), ("reserved_", C.c_int32)]


POSE_RESULT_DTYPE = np.dtype([("E", "<f8", (9,)), ("R", "<f8", (9,)), ("t", "<f8", (3,)), ("n_inliers", "<i4"), ("n_pose_good", "<i4"),
                              ("iters_run", "<i4"), ("n_points", "<i4"), ("n_models", "<i4"), ("reserved_", "<i4")])
assert POSE_RESULT_DTYPE.itemsize == C.sizeof(PoseResult) == 192


class Timings(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_pyramid", C.c_float), ("ms_fast", C.c_float),
                ("ms_select", C.c_float), ("ms_describe", C.c_float), ("ms_knn", C.c_float),
                ("ms_filter", C.c_float), ("ms_pose", C.c_float),
                ("launches_fast", C.c_int32), ("launches_total", C.c_int32), ("ms_update", C.c_float), ("reserved_", C.c_float)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16

SYM_REFERENCE_EFFECTIVE, SYM_INTENDED = 0, 1
STAGE_DETECT, STAGE_MATCH, STAGE_POSE, STAGE_ALL = 1, 2, 4, 7
STAGE_UPDATE, STAGE_FRAME = 8, 15          # Camera::Update's half pyramid at the head of the detect chain; FRAME = ALL | UPDATE
STAGE_GRADIENT = 16                        # Camera::computeGradient into plan-owned buffers, beside the detect chain (implies UPDATE)

# every symbol include/vislam_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "vis_version", "vis_strerror", "vis_device_count", "vis_create", "vis_destroy", "vis_last_error",
    "vis_default_params", "vis_set_params", "vis_get_params", "vis_set_stream", "vis_last_timings",
    "vis_level_geometry", "vis_camera_update", "vis_orb_detect_compute", "vis_bf_knn2_hamming",
    "vis_bf_knn2_hamming_host", "vis_good_matches", "vis_good_matches_host", "vis_essential_ransac",
    "vis_recover_pose", "vis_f2f_ransac", "vis_batch_plan", "vis_batch_reset", "vis_batch_run",
    "vis_batch_sync", "vis_batch_get_keypoints", "vis_batch_get_knn", "vis_batch_get_matches",
    "vis_batch_get_pose", "vis_batch_get_inlier_mask", "vis_debug_counters", "vis_device_pci_bus_id", "vis_batch_status", "vis_synth_canvas", "vis_synth_frame",
    "vis_gradient_frame_elems", "vis_half_pyramid_dims", "vis_gradient_batch", "vis_compute_gradient", "vis_patch_points",
    "vis_image_list", "vis_image_time", "vis_pgm_info", "vis_image_info", "vis_image_read",
    "vis_feeder_create", "vis_feeder_destroy", "vis_feeder_host_buffer", "vis_feeder_submit", "vis_feeder_release",
    "vis_default_align_params", "vis_estimate_pose_features", "vis_align_batch", "vis_batch_align",
    "vis_synth_frame_parallax", "vis_synth_frames_device", "vis_batch_results_async", "vis_batch_half_pyramid", "vis_batch_gradients", "vis_batch_fast_thresholds",
    "vis_se3_exp", "vis_se3_mul", "vis_se3_from_rt", "vis_se3_matrix",
]


def _load():
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(f"libvislam_hip.so not built at {path}: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    lib = C.CDLL(path)
    lib.vis_version.restype = C.c_char_p
    lib.vis_strerror.restype = C.c_char_p
    lib.vis_strerror.argtypes = [C.c_int]
    lib.vis_last_error.restype = C.c_char_p
    lib.vis_last_error.argtypes = [C.c_void_p]
    lib.vis_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.vis_destroy.argtypes = [C.c_void_p]
    lib.vis_destroy.restype = None
    lib.vis_default_params.argtypes = [C.POINTER(Params)]
    lib.vis_default_params.restype = None
    lib.vis_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    lib.vis_get_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    lib.vis_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.vis_last_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    vp, ci, ip = C.c_void_p, C.c_int, C.POINTER(C.c_int)
    lib.vis_level_geometry.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    lib.vis_camera_update.argtypes = [vp, vp, ci, ci, ci, C.POINTER(C.c_void_p)]
    lib.vis_orb_detect_compute.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, ci, ip]
    lib.vis_bf_knn2_hamming.argtypes = [vp, ci, ci, vp, vp]
    lib.vis_bf_knn2_hamming_host.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    lib.vis_good_matches.argtypes = [vp, ci, ci, vp, ci, ip, vp, ci, ip]
    lib.vis_good_matches_host.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, ci, ip, vp, ci, ip]
    lib.vis_essential_ransac.argtypes = [vp, vp, vp, ci, vp, vp, ip, ip]
    lib.vis_recover_pose.argtypes = [vp, vp, vp, vp, ci, vp, vp, ip]
    lib.vis_f2f_ransac.argtypes = [vp, vp, vp, ci, vp, vp, ci, C.c_float, vp, ip]
    lib.vis_batch_plan.argtypes = [vp, ci, ci, ci, ci]
    lib.vis_batch_reset.argtypes = [vp]
    lib.vis_batch_run.argtypes = [vp, vp, ci, ci]
    lib.vis_batch_sync.argtypes = [vp]
    lib.vis_batch_get_keypoints.argtypes = [vp, ci, vp, vp, ci, ip]
    lib.vis_batch_get_knn.argtypes = [vp, ci, vp, ci, ip, vp, ci, ip]
    lib.vis_batch_get_matches.argtypes = [vp, ci, vp, ci, ip, ip]
    lib.vis_batch_get_pose.argtypes = [vp, ci, vp, vp, vp, ip, ip, ip]
    if hasattr(lib, "vis_debug_counters"):              # (absent from older A/B builds selected with VISLAM_HIP_LIB)
        lib.vis_batch_get_inlier_mask.argtypes = [vp, ci, vp, ci, ip]
        lib.vis_debug_counters.argtypes = [vp, vp]
    lib.vis_batch_status.argtypes = [vp, ip]
    lib.vis_synth_canvas.argtypes = [vp, ci, C.c_uint64]
    lib.vis_synth_frame.argtypes = [vp, ci, C.c_uint64, ci, ci, ci, vp, ci]
    lib.vis_gradient_frame_elems.argtypes = [ci, ci]
    lib.vis_gradient_frame_elems.restype = C.c_size_t
    lib.vis_half_pyramid_dims.argtypes = [ci, ci, C.c_void_p, C.c_void_p]
    lib.vis_half_pyramid_dims.restype = None
    lib.vis_gradient_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
    lib.vis_compute_gradient.argtypes = [vp, vp, ci, ci, ci, ci, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.vis_patch_points.argtypes = [vp, vp, ci, ci, C.POINTER(C.c_void_p), ip, C.POINTER(C.c_void_p), ip]
    lib.vis_image_list.argtypes = [C.c_char_p, vp, ci, ip]
    lib.vis_image_time.argtypes = [C.c_char_p]
    lib.vis_image_time.restype = C.c_long
    lib.vis_pgm_info.argtypes = [C.c_char_p, ip, ip]
    lib.vis_image_info.argtypes = [C.c_char_p, ip, ip]
    lib.vis_image_read.argtypes = [C.c_char_p, vp, ci, ci, ci]
    lib.vis_feeder_create.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_void_p)]
    lib.vis_feeder_destroy.argtypes = [vp]
    lib.vis_feeder_destroy.restype = None
    lib.vis_feeder_host_buffer.argtypes = [vp, ci]
    lib.vis_feeder_host_buffer.restype = C.c_void_p
    lib.vis_feeder_submit.argtypes = [vp, ci, ci, C.POINTER(C.c_void_p)]
    lib.vis_feeder_release.argtypes = [vp, ci]
    lib.vis_synth_frame_parallax.argtypes = [vp, ci, C.c_uint64, ci, ci, ci, vp, ci]
    lib.vis_synth_frames_device.argtypes = [vp, vp, ci, C.c_uint64, ci, ci, ci, ci, ci, ci, vp]
    lib.vis_batch_results_async.argtypes = [vp, vp, vp, vp, ci]
    lib.vis_batch_half_pyramid.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.vis_batch_gradients.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.vis_batch_fast_thresholds.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    lib.vis_se3_exp.argtypes = [vp, C.POINTER(Se3f)]; lib.vis_se3_exp.restype = None
    lib.vis_se3_mul.argtypes = [C.POINTER(Se3f), C.POINTER(Se3f), C.POINTER(Se3f)]; lib.vis_se3_mul.restype = None
    lib.vis_se3_from_rt.argtypes = [vp, vp, C.POINTER(Se3f)]; lib.vis_se3_from_rt.restype = None
    lib.vis_se3_matrix.argtypes = [C.POINTER(Se3f), vp]; lib.vis_se3_matrix.restype = None
    lib.vis_default_align_params.argtypes = [C.POINTER(AlignParams)]
    lib.vis_default_align_params.restype = None
    pv = C.POINTER(C.c_void_p)
    lib.vis_estimate_pose_features.argtypes = [vp, C.POINTER(AlignParams), ci, ci, pv, pv, pv, pv, pv, ip, C.POINTER(Se3f), C.POINTER(AlignResult)]
    lib.vis_align_batch.argtypes = [vp, C.POINTER(AlignParams), vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp]
    lib.vis_batch_align.argtypes = [vp, C.POINTER(AlignParams), vp, ci, vp, vp, vp, vp, vp]
    return lib


lib = _load()


def _strerror(code):
    return lib.vis_strerror(int(code)).decode()


def version():
    return lib.vis_version().decode()


def device_pci_bus_id(device):
    """PCI address of a HIP device ("" when the library cannot tell: no device / an older A/B build)"""
    if not hasattr(lib, "vis_device_pci_bus_id"):
        return ""
    buf = C.create_string_buffer(32)
    return buf.value.decode() if lib.vis_device_pci_bus_id(int(device), buf, 32) == 0 else ""


def device_count():
    return int(lib.vis_device_count())


def default_params():
    p = Params()
    lib.vis_default_params(C.byref(p))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_align_params():
    ap = AlignParams()
    lib.vis_default_align_params(C.byref(ap))
    return ap


def se3_exp(a):
    a = np.ascontiguousarray(a, np.float32)
    o = Se3f()
    lib.vis_se3_exp(_ptr(a), C.byref(o))
    return o


def se3_mul(a, b):
    o = Se3f()
    lib.vis_se3_mul(C.byref(a), C.byref(b), C.byref(o))
    return o


def se3_from_rt(R, t):
    R = np.ascontiguousarray(R, np.float32).reshape(9); t = np.ascontiguousarray(t, np.float32)
    o = Se3f()
    lib.vis_se3_from_rt(_ptr(R), _ptr(t), C.byref(o))
    return o


def se3_matrix(a):
    M = np.zeros(16, np.float32)
    lib.vis_se3_matrix(C.byref(a), _ptr(M))
    return M.reshape(4, 4)


# ---- synthetic stream (host-side utility of the library; integer-only, bit-reproducible) ----------
def gradient_frame_elems(w, h):
    return int(lib.vis_gradient_frame_elems(w, h))


def half_pyramid_dims(w, h):
    """(widths, heights) of Camera::Update's 5 levels: cvRound(size * 0.5) per step, round half to even (include/vislam_hip.h)"""
    lw = np.zeros(5, np.int32); lh = np.zeros(5, np.int32)
    lib.vis_half_pyramid_dims(w, h, _ptr(lw), _ptr(lh))
    return [int(x) for x in lw], [int(x) for x in lh]


# -- ImageReader (src/ImageReader.cpp): directory listing, timestamp stems, PGM / raw decode; no GPU involved --------
def image_list(directory):
    n = C.c_int(0)
    rc = lib.vis_image_list(directory.encode(), None, 0, C.byref(n))
    if rc != 0:
        raise VisError(rc, "vis_image_list")
    buf = C.create_string_buffer(max(1, n.value * 300))
    rc = lib.vis_image_list(directory.encode(), buf, len(buf), C.byref(n))
    if rc != 0:
        raise VisError(rc, "vis_image_list")
    return [x for x in buf.value.decode().split("\n") if x]


def image_time(name):
    return int(lib.vis_image_time(name.encode()))


def image_read(path, w=None, h=None):
    if w is None:
        cw, ch = C.c_int(0), C.c_int(0)
        rc = lib.vis_image_info(path.encode(), C.byref(cw), C.byref(ch))
        if rc != 0:
            raise VisError(rc, "vis_image_info")
        w, h = cw.value, ch.value
    out = np.empty((h, w), np.uint8)
    rc = lib.vis_image_read(path.encode(), _ptr(out), w, w, h)
    if rc != 0:
        raise VisError(rc, "vis_image_read")
    return out


class Feeder:
    """pinned-host double-buffered H2D feeder (vis_feeder_*)"""

    def __init__(self, ctx, w, h, batch):
        self.ctx, self.w, self.h, self.batch = ctx, w, h, batch
        self._f = C.c_void_p()
        ctx._chk(lib.vis_feeder_create(ctx._h, w, h, batch, C.byref(self._f)), "vis_feeder_create")

    def host_buffer(self, which):
        p = lib.vis_feeder_host_buffer(self._f, which)
        arr = (C.c_uint8 * (self.batch * self.h * self.w)).from_address(p)
        return np.frombuffer(arr, np.uint8).reshape(self.batch, self.h, self.w)

    def submit(self, which, n):
        d = C.c_void_p()
        self.ctx._chk(lib.vis_feeder_submit(self._f, which, n, C.byref(d)), "vis_feeder_submit")
        return d.value

    def release(self, which):
        self.ctx._chk(lib.vis_feeder_release(self._f, which), "vis_feeder_release")

    def close(self):
        if self._f:
            lib.vis_feeder_destroy(self._f)
            self._f = C.c_void_p()


def synth_canvas(dim=4096, seed=0xE0C00001):
    cv = np.empty((dim, dim), np.uint8)
    rc = lib.vis_synth_canvas(_ptr(cv), dim, C.c_uint64(seed))
    if rc:
        raise VisError(rc, "vis_synth_canvas")
    return cv


def synth_frame(canvas, t, w=752, h=480, seed=0xE0C00001, out=None, parallax=False):
    if out is None:
        out = np.empty((h, w), np.uint8)
    fn = lib.vis_synth_frame_parallax if parallax else lib.vis_synth_frame
    rc = fn(_ptr(canvas), canvas.shape[0], C.c_uint64(seed), int(t), w, h, _ptr(out), out.strides[0])
    if rc:
        raise VisError(rc, "vis_synth_frame")
    return out


class Context:
    """One device context (not thread-safe), like one CameraGPU + MatcherGPU pair."""

    def __init__(self, device=0, params=None):
        self._h = C.c_void_p()
        rc = lib.vis_create(int(device), C.byref(self._h))
        if rc:
            self._h = None
            raise VisError(rc, "vis_create")
        self.params = default_params()
        if params is not None:
            self.set_params(params)

    def close(self):
        if getattr(self, "_h", None):
            lib.vis_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc:
            raise VisError(rc, where, lib.vis_last_error(self._h).decode())

    def set_params(self, p):
        self._chk(lib.vis_set_params(self._h, C.byref(p)), "vis_set_params")
        self.params = p.copy()

    def synth_frames_device(self, d_canvas_ptr, dim, seed, t0, n, w, h, stride, d_out_ptr, parallax=False):
        """frames t0 .. t0+n-1 of the synthetic stream straight into device memory (asynchronous)"""
        self._chk(lib.vis_synth_frames_device(self._h, C.c_void_p(d_canvas_ptr), dim, C.c_uint64(seed), t0, n, w, h, stride,
                                              1 if parallax else 0, C.c_void_p(d_out_ptr)), "vis_synth_frames_device")

    def set_stream(self, raw_stream):
        self._chk(lib.vis_set_stream(self._h, C.c_void_p(raw_stream)), "vis_set_stream")

    def timings(self):
        t = Timings()
        self._chk(lib.vis_last_timings(self._h, C.byref(t)), "vis_last_timings")
        return t

    def level_geometry(self, w, h):
        L = self.params.nlevels
        ws, hs, q = (np.zeros(L, np.int32) for _ in range(3))
        sc = np.zeros(L, np.float32)
        self._chk(lib.vis_level_geometry(self._h, w, h, _ptr(ws), _ptr(hs), _ptr(sc), _ptr(q)), "vis_level_geometry")
        return ws, hs, sc, q

    # -- Camera::Update --------------------------------------------------------------------------------
    def camera_update(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        lw, lh = half_pyramid_dims(w, h)
        levels = [np.empty((lh[l], lw[l]), np.uint8) for l in range(5)]
        arr = (C.c_void_p * 5)(*[l.ctypes.data for l in levels])
        self._chk(lib.vis_camera_update(self._h, _ptr(img), w, h, img.strides[0], arr), "vis_camera_update")
        return levels

    # -- Camera::computeGradient (src/Camera.cpp:167-184) ------------------------------------------------
    def compute_gradient(self, img, scale=3):
        """one host frame -> per level (gx int16, gy int16, gradient u8) of the 5 half-pyramid levels"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        lw, lh = half_pyramid_dims(w, h)
        gx = [np.empty((lh[l], lw[l]), np.int16) for l in range(5)]
        gy = [np.empty((lh[l], lw[l]), np.int16) for l in range(5)]
        g = [np.empty((lh[l], lw[l]), np.uint8) for l in range(5)]
        ax = (C.c_void_p * 5)(*[a.ctypes.data for a in gx])
        ay = (C.c_void_p * 5)(*[a.ctypes.data for a in gy])
        ag = (C.c_void_p * 5)(*[a.ctypes.data for a in g])
        self._chk(lib.vis_compute_gradient(self._h, _ptr(img), w, h, img.strides[0], scale, ax, ay, ag), "vis_compute_gradient")
        return gx, gy, g

    def gradient_batch(self, d_frames_ptr, w, h, stride, n, d_gray_ptr, d_gx_ptr, d_gy_ptr, d_g_ptr, scale=3):
        """device pointers in, device buffers out (n * gradient_frame_elems(w, h) elements each); asynchronous"""
        self._chk(lib.vis_gradient_batch(self._h, C.c_void_p(d_frames_ptr), w, h, stride, n, scale, C.c_void_p(d_gray_ptr),
                                         C.c_void_p(d_gx_ptr), C.c_void_p(d_gy_ptr), C.c_void_p(d_g_ptr)), "vis_gradient_batch")

    # -- Camera::ObtainPatchesPointsPreviousFrame / ObtainDebugPointsPreviousFrame (src/Camera.cpp:358-445) --
    def patch_points(self, good, cap=200 * 121):
        good = np.ascontiguousarray(good, KEYPOINT_DTYPE)
        patch = [np.zeros((cap, 4), np.float32) for _ in range(5)]
        debug = [np.zeros((cap, 4), np.float32) for _ in range(5)]
        ap = (C.c_void_p * 5)(*[a.ctypes.data for a in patch])
        ad = (C.c_void_p * 5)(*[a.ctypes.data for a in debug])
        npt = (C.c_int * 5)(); ndb = (C.c_int * 5)()
        self._chk(lib.vis_patch_points(self._h, _ptr(good), len(good), cap, ap, npt, ad, ndb), "vis_patch_points")
        return [patch[l][:npt[l]].copy() for l in range(5)], [debug[l][:ndb[l]].copy() for l in range(5)]

    # -- VISystem::EstimatePoseFeatures (src/VISystem.cpp:1113-1448) ------------------------------------------
    def estimate_pose_features(self, ap, w, h, gray1, gray2, gx1, gy1, cand1, init=None):
        """one pair, host arrays: per-level lists (None for unused levels) -> AlignResult"""
        def lv(arrs, dt):
            keep = [None if a is None else np.ascontiguousarray(a, dt) for a in arrs]
            keep += [None] * (5 - len(keep))
            return keep, (C.c_void_p * 5)(*[None if a is None else a.ctypes.data for a in keep])
        k1, a1 = lv(gray1, np.uint8); k2, a2 = lv(gray2, np.uint8); k3, a3 = lv(gx1, np.int16); k4, a4 = lv(gy1, np.int16)
        k5, a5 = lv(cand1, np.float32)
        n = (C.c_int * 5)(*[0 if c is None else len(c) for c in k5])
        res = AlignResult()
        self._chk(lib.vis_estimate_pose_features(self._h, C.byref(ap), w, h, a1, a2, a3, a4, a5, n,
                                                 None if init is None else C.byref(init), C.byref(res)), "vis_estimate_pose_features")
        return res

    def align_batch(self, ap, d_frames_ptr, w, h, stride, n, d_gray_ptr, d_gx_ptr, d_gy_ptr, d_pts_ptr, d_npts_ptr, max_pts,
                    d_init_ptr, d_out_ptr):
        """device pointers; asynchronous on the context's stream"""
        self._chk(lib.vis_align_batch(self._h, C.byref(ap), C.c_void_p(d_frames_ptr), w, h, stride, n, C.c_void_p(d_gray_ptr),
                                      C.c_void_p(d_gx_ptr), C.c_void_p(d_gy_ptr), C.c_void_p(d_pts_ptr), C.c_void_p(d_npts_ptr), max_pts,
                                      C.c_void_p(d_init_ptr) if d_init_ptr else None, C.c_void_p(d_out_ptr)), "vis_align_batch")

    def batch_align(self, ap, d_frames_ptr, n, d_gray_ptr, d_gx_ptr, d_gy_ptr, d_init_ptr, d_out_ptr):
        """alignment of the pairs of the last batch_run, matched points taken from the plan; asynchronous"""
        nz = lambda q: C.c_void_p(q) if q else None                 # 0 = NULL: gradient pointers all NULL -> the plan's (STAGE_GRADIENT)
        self._chk(lib.vis_batch_align(self._h, C.byref(ap), C.c_void_p(d_frames_ptr), n, nz(d_gray_ptr), nz(d_gx_ptr),
                                      nz(d_gy_ptr), nz(d_init_ptr), C.c_void_p(d_out_ptr)),
                  "vis_batch_align")

    # -- CameraGPU::detectAndComputeGPUFeatures ---------------------------------------------------------
    def orb_detect_compute(self, img, slot=0, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        if cap is None:
            cap = 2 * self.params.nfeatures + 1024
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._chk(lib.vis_orb_detect_compute(self._h, _ptr(img), w, h, img.strides[0], slot, _ptr(kps), _ptr(desc), cap,
                                             C.byref(n)), "vis_orb_detect_compute")
        return kps[:n.value].copy(), desc[:n.value].copy()

    # -- MatcherGPU::computeGPUMatches ---------------------------------------------------------------------
    def bf_knn2_hamming(self, slot_q, slot_t, nq, nt):
        o12 = np.zeros((nq, 2), DMATCH_DTYPE)
        o21 = np.zeros((nt, 2), DMATCH_DTYPE)
        self._chk(lib.vis_bf_knn2_hamming(self._h, slot_q, slot_t, _ptr(o12), _ptr(o21)), "vis_bf_knn2_hamming")
        return o12, o21

    def bf_knn2_hamming_host(self, d1, d2):
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        o12 = np.zeros((len(d1), 2), DMATCH_DTYPE)
        o21 = np.zeros((len(d2), 2), DMATCH_DTYPE)
        self._chk(lib.vis_bf_knn2_hamming_host(self._h, _ptr(d1), len(d1), _ptr(d2), len(d2), _ptr(o12), _ptr(o21)),
                  "vis_bf_knn2_hamming_host")
        return o12, o21

    # -- Matcher::computeBestMatches ----------------------------------------------------------------------
    def good_matches(self, slot_prev, slot_cur, sym_cap=65536):
        root2 = 1024
        good = np.zeros(root2, DMATCH_DTYPE)
        sym = np.zeros(sym_cap, DMATCH_DTYPE)
        ng, ns = C.c_int(0), C.c_int(0)
        self._chk(lib.vis_good_matches(self._h, slot_prev, slot_cur, _ptr(good), root2, C.byref(ng), _ptr(sym), sym_cap,
                                       C.byref(ns)), "vis_good_matches")
        return good[:ng.value].copy(), sym[:ns.value].copy()

    def good_matches_host(self, kps1, kps2, knn12, knn21):
        kps1 = np.ascontiguousarray(kps1, KEYPOINT_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KEYPOINT_DTYPE)
        knn12 = np.ascontiguousarray(knn12, DMATCH_DTYPE)
        knn21 = np.ascontiguousarray(knn21, DMATCH_DTYPE)
        good = np.zeros(1024, DMATCH_DTYPE)
        sym = np.zeros(max(len(kps1), 1), DMATCH_DTYPE)
        ng, ns = C.c_int(0), C.c_int(0)
        self._chk(lib.vis_good_matches_host(self._h, _ptr(kps1), len(kps1), _ptr(kps2), len(kps2), _ptr(knn12), _ptr(knn21),
                                            _ptr(good), 1024, C.byref(ng), _ptr(sym), len(sym), C.byref(ns)),
                  "vis_good_matches_host")
        return good[:ng.value].copy(), sym[:ns.value].copy()

    # -- findEssentialMat / recoverPose --------------------------------------------------------------------
    def essential_ransac(self, p1, p2):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        E = np.zeros(9, np.float64)
        mask = np.zeros(max(len(p1), 1), np.uint8)
        ni, it = C.c_int(0), C.c_int(0)
        self._chk(lib.vis_essential_ransac(self._h, _ptr(p1), _ptr(p2), len(p1), _ptr(E), _ptr(mask), C.byref(ni), C.byref(it)),
                  "vis_essential_ransac")
        return E.reshape(3, 3), mask[:len(p1)], ni.value, it.value

    def recover_pose(self, E, p1, p2):
        E = np.ascontiguousarray(E, np.float64).reshape(9)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        R = np.zeros(9, np.float64)
        t = np.zeros(3, np.float64)
        ng = C.c_int(0)
        self._chk(lib.vis_recover_pose(self._h, _ptr(E), _ptr(p1), _ptr(p2), len(p1), _ptr(R), _ptr(t), C.byref(ng)),
                  "vis_recover_pose")
        return R.reshape(3, 3), t, ng.value

    def f2f_ransac(self, pts1, pts2, rot, sample_idx, scale):
        pts1 = np.ascontiguousarray(pts1, KEYPOINT_DTYPE)
        pts2 = np.ascontiguousarray(pts2, KEYPOINT_DTYPE)
        rot = np.ascontiguousarray(rot, np.float32).reshape(9)
        idx = np.ascontiguousarray(sample_idx, np.int32).reshape(-1)
        out = np.zeros(3, np.float32)
        cm = C.c_int(0)
        self._chk(lib.vis_f2f_ransac(self._h, _ptr(pts1), _ptr(pts2), len(pts1), _ptr(rot), _ptr(idx), len(idx) // 2,
                                     C.c_float(scale), _ptr(out), C.byref(cm)), "vis_f2f_ransac")
        return out, cm.value

    # -- batched stream path ----------------------------------------------------------------------------------
    def batch_plan(self, w, h, stride, max_frames):
        self._chk(lib.vis_batch_plan(self._h, w, h, stride, max_frames), "vis_batch_plan")

    def batch_reset(self):
        self._chk(lib.vis_batch_reset(self._h), "vis_batch_reset")

    def batch_run(self, dev_ptr, n_frames, stages=STAGE_ALL):
        self._chk(lib.vis_batch_run(self._h, C.c_void_p(dev_ptr), n_frames, stages), "vis_batch_run")

    def batch_sync(self):
        self._chk(lib.vis_batch_sync(self._h), "vis_batch_sync")

    def batch_results_async(self, n_cap, h_pose_ptr=None, h_good_ptr=None, h_ngood_ptr=None):
        """queue the D2H copy of the last batch's results (raw host pointers, ideally pinned, holding n_cap frames); the host may
        read them after batch_sync()"""
        self._chk(lib.vis_batch_results_async(self._h, C.c_void_p(h_pose_ptr) if h_pose_ptr else None,
                                              C.c_void_p(h_good_ptr) if h_good_ptr else None,
                                              C.c_void_p(h_ngood_ptr) if h_ngood_ptr else None, n_cap), "vis_batch_results_async")

    def batch_half_pyramid(self):
        """(device pointer, frame_elems) of the half pyramids the last batch_run(..., STAGE_UPDATE) wrote"""
        d, fe = C.c_void_p(0), C.c_size_t(0)
        self._chk(lib.vis_batch_half_pyramid(self._h, C.byref(d), C.byref(fe)), "vis_batch_half_pyramid")
        return d.value, fe.value

    def batch_gradients(self):
        """(gray, gx, gy, g device pointers, frame_elems) of the gradients the last batch_run(..., STAGE_GRADIENT) wrote"""
        p = [C.c_void_p() for _ in range(4)]; fe = C.c_size_t()
        self._chk(lib.vis_batch_gradients(self._h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(p[3]), C.byref(fe)), "vis_batch_gradients")
        return p[0].value, p[1].value, p[2].value, p[3].value, fe.value

    def batch_results(self, n):
        """synchronous convenience: (pose records, good matches n x root^2, counts) of the last batch"""
        root2 = int(np.floor(np.sqrt(self.params.n_cells))) ** 2
        pose = np.zeros(n, POSE_RESULT_DTYPE); good = np.zeros((n, root2), DMATCH_DTYPE); ng = np.zeros(n, np.int32)
        self.batch_results_async(n, pose.ctypes.data, good.ctypes.data, ng.ctypes.data)
        self.batch_sync()
        return pose, good, ng

    def batch_fast_thresholds(self):
        """(per-level FAST thresholds the next batch starts from, (frame, level) pairs the last batch had to redo)"""
        tau = np.zeros(self.params.nlevels, np.int32)
        nr = C.c_int32(0)
        self._chk(lib.vis_batch_fast_thresholds(self._h, _ptr(tau), C.byref(nr)), "vis_batch_fast_thresholds")
        return tau, nr.value

    def batch_status(self):
        f = C.c_int(0)
        self._chk(lib.vis_batch_status(self._h, C.byref(f)), "vis_batch_status")
        return f.value

    def batch_keypoints(self, frame, cap=None):
        if cap is None:
            cap = 2 * self.params.nfeatures + 1024
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._chk(lib.vis_batch_get_keypoints(self._h, frame, _ptr(kps), _ptr(desc), cap, C.byref(n)), "vis_batch_get_keypoints")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def batch_knn(self, frame, cap=None):
        if cap is None:
            cap = 2 * self.params.nfeatures + 1024
        o12 = np.zeros((cap, 2), DMATCH_DTYPE)
        o21 = np.zeros((cap, 2), DMATCH_DTYPE)
        n12, n21 = C.c_int(0), C.c_int(0)
        self._chk(lib.vis_batch_get_knn(self._h, frame, _ptr(o12), 2 * cap, C.byref(n12), _ptr(o21), 2 * cap, C.byref(n21)),
                  "vis_batch_get_knn")
        return o12[:n12.value].copy(), o21[:n21.value].copy()

    def batch_matches(self, frame):
        good = np.zeros(1024, DMATCH_DTYPE)
        ng, ns = C.c_int(0), C.c_int(0)
        self._chk(lib.vis_batch_get_matches(self._h, frame, _ptr(good), 1024, C.byref(ng), C.byref(ns)), "vis_batch_get_matches")
        return good[:ng.value].copy(), ns.value

    def debug_counters(self):
        """(kernel launches of the process, host waits of this context's single-frame entry points, asynchronous copies they queued)"""
        out = (C.c_ulonglong * 4)()
        self._chk(lib.vis_debug_counters(self._h, out), "vis_debug_counters")
        return int(out[0]), int(out[1]), int(out[2])

    def batch_inlier_mask(self, frame, cap=16384):
        mask = np.zeros(cap, np.uint8)
        n = C.c_int(0)
        self._chk(lib.vis_batch_get_inlier_mask(self._h, frame, _ptr(mask), cap, C.byref(n)), "vis_batch_get_inlier_mask")
        return mask[:n.value].copy()

    def batch_pose(self, frame):
        E, R, t = np.zeros(9), np.zeros(9), np.zeros(3)
        ni, ng, it = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(lib.vis_batch_get_pose(self._h, frame, _ptr(E), _ptr(R), _ptr(t), C.byref(ni), C.byref(ng), C.byref(it)),
                  "vis_batch_get_pose")
        return dict(E=E.reshape(3, 3), R=R.reshape(3, 3), t=t, n_inliers=ni.value, n_pose_good=ng.value, iters_run=it.value)
