"""ctypes binding of liborbhip.so (include/orb_hip.h).

Host-array entry points take numpy arrays; the *_device entry points take
torch CUDA tensors (torch is only the owner of device memory and streams).
There is no CPU fallback: if the library cannot be loaded, importing the
compute classes fails with the loader's error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liborbhip.so")

ORB_Q_RESIZE_AREA, ORB_Q_UMAX_FORK, ORB_Q_DIVIDE_N3, ORB_Q_GRID_EXCLUSIVE, ORB_Q_WINDOW_CIRCULAR, ORB_Q_GAUSS_LEGACY = (
    1 << i for i in range(6)
)
ORB_QUIRKS_FORK = 0x1F
ORB_BF_NOALLOC_MAX_QUERIES = 262144  # include/orb_hip.h
ORB_QUIRKS_CANONICAL = 0

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)

EXPORTS = [
    "orb_create", "orb_destroy", "orb_max_keypoints_per_frame", "orb_extract", "orb_extract_batch",
    "orb_extract_batch_device", "orb_check_last_batch", "orb_quadtree_second_pass_levels", "orb_get_pyramid_level", "orb_get_pyramid", "orb_get_tables",
    "orb_compute_tables",
    "orb_debug_get_fast_keys", "orb_resize", "orb_gaussian_blur", "orb_hamming", "orb_match_window",
    "orb_match_window_batch_device", "orb_match_reserve_candidates", "orb_match_projected", "orb_bf_create", "orb_bf_destroy", "orb_bf_set_database",
    "orb_bf_set_database_device", "orb_bf_query_top2", "orb_bf_query_packed_device", "orb_bf_query_packed_noalloc", "orb_bf_clock_probe", "orb_bf_merge_packed_device",
    "orb_bf_multi_create", "orb_bf_multi_destroy", "orb_bf_multi_exchange", "orb_bf_multi_set_database",
    "orb_bf_multi_query_top2", "orb_extract_batch_multi", "orb_extract_batch_multi_device", "orb_alloc_host", "orb_free_host",
    "orb_enable_stage_timing", "orb_get_stage_timings", "orb_last_error", "orb_version",
    "orb_distinctive_descriptors", "orb_distinctive_descriptors_device",
    "orb_undistort_keypoints", "orb_undistort_keypoints_device", "orb_extract_undistorted", "orb_compute_image_bounds", "orb_set_image_bounds",
    "orb_voc_load_text", "orb_voc_create", "orb_voc_destroy", "orb_voc_info", "orb_voc_transform",
    "orb_voc_transform_device", "orb_voc_bow", "orb_bow_score_l1", "orb_match_bow",
    "orb_match_local_map", "orb_match_best", "orb_match_bow_keyframes", "orb_match_triangulation",
]


class OrbParams(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int), ("ini_th_fast", C.c_int),
        ("min_th_fast", C.c_int), ("max_width", C.c_int), ("max_height", C.c_int), ("max_batch", C.c_int),
        ("device", C.c_int), ("quirks", C.c_uint),
    ]


class OrbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("liborbhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C orb_slam2_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP/HSA runtime under the same soname.  Load it first so that
        # this process holds ONE runtime: device pointers and streams of torch tensors are then valid
        # inside liborbhip, and a second runtime never fights the first for the device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.orb_last_error.restype = C.c_char_p
        L.orb_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise OrbError(rc, lib().orb_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(stream):
    if stream is None:
        import torch

        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def hamming(a, b):
    """ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:460-476)."""
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orb_hamming(_p(a), _p(b))


class Context:
    """One ORBextractor (+ windowed matcher state) on one device."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, max_width=640, max_height=480,
                 max_batch=1, device=0, quirks=ORB_QUIRKS_FORK):
        self.params = OrbParams(nfeatures, scale, nlevels, ini_th, min_th, max_width, max_height, max_batch, device, quirks)
        self.h = C.c_void_p()
        self.nlevels = nlevels
        self._destroy = lib().orb_destroy  # kept on the object: module globals may be gone at interpreter exit
        _check(lib().orb_create(C.byref(self.params), C.byref(self.h)))
        self.cap = lib().orb_max_keypoints_per_frame(self.h)

    def close(self):
        if getattr(self, "h", None) and self.h:
            self._destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    # ---- extraction, host arrays -----------------------------------------------------
    def extract(self, img):
        """ORBextractor::operator()(image, mask, keypoints, descriptors)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape if img.ndim == 2 else (0, 0)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        _check(lib().orb_extract(self.h, _p(img), w, h, img.strides[0] if img.ndim == 2 and h else w, _p(kps), _p(desc), self.cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch(self, imgs):
        imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        n = len(imgs)
        h, w = imgs[0].shape
        assert all(i.shape == (h, w) for i in imgs)
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        kps = np.zeros((n, self.cap), KP_DTYPE)
        desc = np.zeros((n, self.cap, 32), np.uint8)
        cnt = np.zeros(n, np.int32)
        _check(lib().orb_extract_batch(self.h, ptrs, n, w, h, imgs[0].strides[0], _p(kps), _p(desc), self.cap, _p(cnt)))
        return [(kps[f, : cnt[f]].copy(), desc[f, : cnt[f]].copy()) for f in range(n)]

    def extract_batch_into(self, imgs, kps, desc, cnt):
        """orb_extract_batch into arrays the caller keeps (kps [n, cap] KP_DTYPE, desc [n, cap, 32] uint8, cnt [n]
        int32): nothing is allocated per call, which is half the time of a 64-frame call otherwise.  Returns cnt."""
        n = len(imgs)
        h, w = imgs[0].shape
        assert all(i.shape == (h, w) and i.dtype == np.uint8 and i.strides == imgs[0].strides for i in imgs)
        assert kps.shape == (n, self.cap) and desc.shape == (n, self.cap, 32) and cnt.shape == (n,)
        assert kps.flags.c_contiguous and desc.flags.c_contiguous and cnt.dtype == np.int32
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        _check(lib().orb_extract_batch(self.h, ptrs, n, w, h, imgs[0].strides[0], _p(kps), _p(desc), self.cap, _p(cnt)))
        return cnt

    # ---- extraction, device tensors ----------------------------------------------------
    def extract_batch_device(self, d_imgs, d_kps, d_desc, d_counts, stream=None):
        """d_imgs: uint8 CUDA tensor [n, h, w] (any row / frame stride); outputs are
        CUDA tensors: d_kps uint8 [n, cap, 28], d_desc uint8 [n, cap, 32], d_counts int32 [n]."""
        n, h, w = d_imgs.shape
        assert d_imgs.stride(2) == 1
        cap = d_kps.shape[1]
        _check(lib().orb_extract_batch_device(
            self.h, _dp(d_imgs), n, w, h, d_imgs.stride(1), C.c_int64(d_imgs.stride(0)), _dp(d_kps), _dp(d_desc), cap,
            _dp(d_counts), _stream(stream)))

    def enable_stage_timing(self, on=True):
        _check(lib().orb_enable_stage_timing(self.h, int(on)))

    def stage_timings(self):
        """Mean ms per batch of (pyramid, fast, quadtree, describe) since the last call, and the batch count."""
        ms = (C.c_float * 4)()
        n = C.c_int(0)
        _check(lib().orb_get_stage_timings(self.h, ms, C.byref(n)))
        return dict(pyramid=ms[0], fast=ms[1], quadtree=ms[2], describe=ms[3]), n.value

    def quadtree_second_pass_levels(self):
        n = C.c_int(0)
        _check(lib().orb_quadtree_second_pass_levels(self.h, C.byref(n)))
        return n.value

    def check_last_batch(self):
        _check(lib().orb_check_last_batch(self.h))

    # ---- stage access ---------------------------------------------------------------------
    def tables(self):
        L = self.nlevels
        f = [np.zeros(L, np.float32) for _ in range(4)]
        per = np.zeros(L, np.int32)
        um = np.zeros(16, np.int32)
        _check(lib().orb_get_tables(self.h, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(per), _p(um)))
        return dict(scale=f[0], inv_scale=f[1], sigma2=f[2], inv_sigma2=f[3], per_level=per, umax=um)

    def pyramid_level(self, frame, level):
        """mvImagePyramid[level] of a frame of the last batch."""
        w, h = C.c_int(), C.c_int()
        _check(lib().orb_get_pyramid_level(self.h, frame, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _check(lib().orb_get_pyramid_level(self.h, frame, level, _p(out), w.value, C.byref(w), C.byref(h)))
        return out

    def fast_keys(self, frame, level):
        n = C.c_int(0)
        _check(lib().orb_debug_get_fast_keys(self.h, frame, level, None, None, None, 0, C.byref(n)))
        xs, ys, sc = (np.zeros(max(n.value, 1), np.int32) for _ in range(3))
        _check(lib().orb_debug_get_fast_keys(self.h, frame, level, _p(xs), _p(ys), _p(sc), n.value, C.byref(n)))
        return xs[: n.value], ys[: n.value], sc[: n.value]

    def resize(self, src, dw, dh, area=True):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        _check(lib().orb_resize(self.h, _p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw, int(area)))
        return dst

    def gaussian_blur(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros_like(src)
        _check(lib().orb_gaussian_blur(self.h, _p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0]))
        return dst

    # ---- matching ---------------------------------------------------------------------------
    def match_window(self, k1, d1, k2, d2, img_w, img_h, prev_xy, window=100, nnratio=0.9, check_ori=True):
        """ORBmatcher::SearchForInitialization.  Returns (nmatches, matches12, prev_xy_out)."""
        k1 = np.ascontiguousarray(k1, KP_DTYPE)
        k2 = np.ascontiguousarray(k2, KP_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8)
        d2 = np.ascontiguousarray(d2, np.uint8)
        prev = np.ascontiguousarray(prev_xy, np.float32).copy()
        m12 = np.zeros(max(len(k1), 1), np.int32)
        nm = C.c_int(0)
        _check(lib().orb_match_window(self.h, _p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), img_w, img_h, _p(prev),
                                      _p(m12), window, C.c_float(nnratio), int(check_ori), C.byref(nm)))
        return nm.value, m12[: len(k1)], prev

    def match_window_batch_device(self, d_k1, d_d1, d_n1, d_k2, d_d2, d_n2, img_w, img_h, d_prev, d_m12, d_nm,
                                  window=100, nnratio=0.9, check_ori=True, stream=None):
        npairs, s1 = d_k1.shape[0], d_k1.shape[1]
        s2 = d_k2.shape[1]
        _check(lib().orb_match_window_batch_device(
            self.h, npairs, _dp(d_k1), _dp(d_d1), _dp(d_n1), s1, _dp(d_k2), _dp(d_d2), _dp(d_n2), s2, img_w, img_h,
            _dp(d_prev), _dp(d_m12), window, C.c_float(nnratio), int(check_ori), _dp(d_nm), _stream(stream)))

    def match_projected(self, u, v, radius, min_level, max_level, qangle, qdesc, qobs, k2, d2, img_w, img_h,
                        occupied2=None, th_high=100, check_ori=True):
        """Search loop of ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        u, v, radius, qangle = f32(u), f32(v), f32(radius), f32(qangle)
        min_level, max_level = i32(min_level), i32(max_level)
        qdesc, d2, qobs, occ = u8(qdesc), u8(d2), u8(qobs), u8(occupied2)
        k2 = np.ascontiguousarray(k2, KP_DTYPE)
        match2 = np.zeros(max(len(k2), 1), np.int32)
        nm = C.c_int(0)
        _check(lib().orb_match_projected(self.h, _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(qangle),
                                         _p(qdesc), _p(qobs), len(u), _p(k2), _p(d2), len(k2), img_w, img_h, _p(occ),
                                         _p(match2), th_high, int(check_ori), C.byref(nm)))
        return nm.value, match2[: len(k2)]

    def match_local_map(self, u, v, radius, min_level, max_level, qdesc, qobs, k2, d2, img_w, img_h, occupied2=None,
                        th_high=100, nnratio=0.8):
        """Search loop of ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
        (include/ORBmatcher.h:24-39).  Returns (nmatches, match2[n2])."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        u, v, radius = f32(u), f32(v), f32(radius)
        min_level, max_level = i32(min_level), i32(max_level)
        qdesc, d2, qobs, occ = u8(qdesc), u8(d2), u8(qobs), u8(occupied2)
        k2 = np.ascontiguousarray(k2, KP_DTYPE)
        match2 = np.zeros(max(len(k2), 1), np.int32)
        nm = C.c_int(0)
        _check(lib().orb_match_local_map(self.h, _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(qdesc),
                                         _p(qobs), len(u), _p(k2), _p(d2), len(k2), img_w, img_h, _p(occ), _p(match2),
                                         th_high, C.c_float(nnratio), C.byref(nm)))
        return nm.value, match2[: len(k2)]

    def match_best(self, u, v, radius, min_level, max_level, qdesc, k2, d2, img_w, img_h, inv_level_sigma2=None,
                   chi2_max=5.99, th=50):
        """Candidate loop of ORBmatcher::Fuse / SearchBySim3 (include/ORBmatcher.h:146-184): the best keypoint
        of every query on its own.  Returns (best_idx, best_dist) int32 arrays."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        u, v, radius = f32(u), f32(v), f32(radius)
        min_level, max_level = i32(min_level), i32(max_level)
        qdesc, d2 = np.ascontiguousarray(qdesc, np.uint8), np.ascontiguousarray(d2, np.uint8)
        k2 = np.ascontiguousarray(k2, KP_DTYPE)
        sig = None if inv_level_sigma2 is None else f32(inv_level_sigma2)
        idx = np.zeros(max(len(u), 1), np.int32)
        dist = np.zeros(max(len(u), 1), np.int32)
        _check(lib().orb_match_best(self.h, _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(qdesc), len(u),
                                    _p(k2), _p(d2), len(k2), img_w, img_h, _p(sig), 0 if sig is None else len(sig),
                                    C.c_double(chi2_max), th, _p(idx), _p(dist)))
        return idx[: len(u)], dist[: len(u)]

    def undistort_keypoints(self, kps, K, dist):
        """Frame::UndistortKeyPoints; K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2, k3)."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        out = np.zeros(len(kps), KP_DTYPE)
        K = np.ascontiguousarray(K, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        _check(lib().orb_undistort_keypoints(self.h, _p(K), _p(dist), _p(kps), _p(out), len(kps)))
        return out

    def extract_undistorted(self, img, K, dist):
        """ExtractORB + UndistortKeyPoints of the Frame constructor for one host frame in one call:
        (keypoints, undistorted keypoints, descriptors)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        K = np.ascontiguousarray(K, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        kps = np.zeros(self.cap, KP_DTYPE)
        un = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        _check(lib().orb_extract_undistorted(self.h, _p(img), w, h, img.strides[0], _p(K), _p(dist), _p(kps), _p(un), _p(desc),
                                             self.cap, C.byref(n)))
        return kps[: n.value], un[: n.value], desc[: n.value]

    def undistort_keypoints_device(self, d_in, d_out, n, K, dist, stream=None):
        K = np.ascontiguousarray(K, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        _check(lib().orb_undistort_keypoints_device(self.h, _p(K), _p(dist), _dp(d_in), _dp(d_out), C.c_longlong(n), _stream(stream)))

    def compute_image_bounds(self, cols, rows, K, dist):
        """Frame::ComputeImageBounds; the context keeps the bounds for the windowed matchers."""
        K = np.ascontiguousarray(K, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        b = np.zeros(4, np.float32)
        _check(lib().orb_compute_image_bounds(self.h, cols, rows, _p(K), _p(dist), _p(b)))
        return b

    def set_image_bounds(self, bounds):
        b = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
        _check(lib().orb_set_image_bounds(self.h, _p(b)))

    def distinctive_descriptors(self, desc, offsets, self_distance=False):
        """MapPoint::ComputeDistinctiveDescriptors for the map points whose observation descriptors are
        desc[offsets[p]:offsets[p+1]].  Returns (best_index, best_median) int32 arrays."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        offsets = np.ascontiguousarray(offsets, np.int32)
        npts = len(offsets) - 1
        best = np.zeros(max(npts, 1), np.int32)
        med = np.zeros(max(npts, 1), np.int32)
        _check(lib().orb_distinctive_descriptors(self.h, _p(desc), _p(offsets), npts, int(self_distance), _p(best), _p(med)))
        return best[:npts], med[:npts]

    def distinctive_descriptors_device(self, d_desc, d_offsets, d_best, d_median=None, self_distance=False, stream=None):
        _check(lib().orb_distinctive_descriptors_device(self.h, _dp(d_desc), _dp(d_offsets), d_offsets.shape[0] - 1,
                                                        int(self_distance), _dp(d_best),
                                                        _dp(d_median) if d_median is not None else None, _stream(stream)))


    def match_bow(self, kf_keys_un, kf_desc, kf_valid, kf_fv, f_keys, f_desc, f_fv, nnratio=0.7, check_ori=True):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:138-268).  kf_fv / f_fv are
        feature vectors (nodes, offsets, indices) as Vocabulary.transform returns them; kf_valid marks the
        keyframe features that hold a usable map point.  Returns (nmatches, match_kf[nf])."""
        kk = np.ascontiguousarray(kf_keys_un, KP_DTYPE)
        fk = np.ascontiguousarray(f_keys, KP_DTYPE)
        kd = np.ascontiguousarray(kf_desc, np.uint8)
        fd = np.ascontiguousarray(f_desc, np.uint8)
        kv = np.ascontiguousarray(kf_valid, np.uint8)
        a = [np.ascontiguousarray(x, np.int32) for x in kf_fv]
        b = [np.ascontiguousarray(x, np.int32) for x in f_fv]
        match = np.full(max(len(fk), 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().orb_match_bow(self.h, _p(kk), _p(kd), _p(kv), len(kk), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]), _p(fk),
                                   _p(fd), len(fk), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]), C.c_float(nnratio),
                                   int(check_ori), _p(match), C.byref(nm)))
        return nm.value, match[: len(fk)]

    def match_bow_keyframes(self, keys_un1, desc1, valid1, fv1, keys_un2, desc2, valid2, fv2, nnratio=0.75, check_ori=True):
        """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (include/ORBmatcher.h:110).  Returns
        (nmatches, matches12[n1])."""
        k1 = np.ascontiguousarray(keys_un1, KP_DTYPE)
        k2 = np.ascontiguousarray(keys_un2, KP_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        v1, v2 = np.ascontiguousarray(valid1, np.uint8), np.ascontiguousarray(valid2, np.uint8)
        a = [np.ascontiguousarray(x, np.int32) for x in fv1]
        b = [np.ascontiguousarray(x, np.int32) for x in fv2]
        match = np.full(max(len(k1), 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().orb_match_bow_keyframes(self.h, _p(k1), _p(d1), _p(v1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]),
                                             len(a[0]), _p(k2), _p(d2), _p(v2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]),
                                             len(b[0]), C.c_float(nnratio), int(check_ori), _p(match), C.byref(nm)))
        return nm.value, match[: len(k1)]

    def match_triangulation(self, keys_un1, desc1, free1, fv1, keys_un2, desc2, free2, fv2, F12, epipole, scale_factors,
                            level_sigma2, mark_matched=False, check_ori=True):
        """ORBmatcher::SearchForTriangulation (include/ORBmatcher.h:133-144), monocular.  Returns
        (nmatches, matches12[n1])."""
        k1 = np.ascontiguousarray(keys_un1, KP_DTYPE)
        k2 = np.ascontiguousarray(keys_un2, KP_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        f1, f2 = np.ascontiguousarray(free1, np.uint8), np.ascontiguousarray(free2, np.uint8)
        a = [np.ascontiguousarray(x, np.int32) for x in fv1]
        b = [np.ascontiguousarray(x, np.int32) for x in fv2]
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        ep = None if epipole is None else np.ascontiguousarray(epipole, np.float32)
        sc = np.ascontiguousarray(scale_factors, np.float32)
        sg = np.ascontiguousarray(level_sigma2, np.float32)
        match = np.full(max(len(k1), 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().orb_match_triangulation(self.h, _p(k1), _p(d1), _p(f1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]),
                                             len(a[0]), _p(k2), _p(d2), _p(f2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]),
                                             len(b[0]), _p(F), _p(ep), _p(sc), _p(sg), len(sc), int(mark_matched),
                                             int(check_ori), _p(match), C.byref(nm)))
        return nm.value, match[: len(k1)]


class BruteForce:
    """One shard of a brute-force descriptor database on one device."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        self.device = device
        self._destroy = lib().orb_bf_destroy
        _check(lib().orb_bf_create(device, C.byref(self.h)))
        self._keep = None

    def close(self):
        if getattr(self, "h", None) and self.h:
            self._destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def set_database(self, db, global_base=0):
        db = np.ascontiguousarray(db, np.uint8)
        _check(lib().orb_bf_set_database(self.h, _p(db), C.c_int64(len(db)), C.c_int64(global_base)))

    def set_database_device(self, d_db, global_base=0):
        self._keep = d_db  # the shard keeps the pointer, not a copy
        _check(lib().orb_bf_set_database_device(self.h, _dp(d_db), C.c_int64(d_db.shape[0]), C.c_int64(global_base)))

    def query_top2(self, q):
        q = np.ascontiguousarray(q, np.uint8)
        nq = len(q)
        idx = np.zeros(nq, np.int32)
        d1 = np.zeros(nq, np.uint16)
        d2 = np.zeros(nq, np.uint16)
        _check(lib().orb_bf_query_top2(self.h, _p(q), nq, _p(idx), _p(d1), _p(d2)))
        return idx, d1, d2

    def query_packed_device(self, d_q, d_packed, stream=None):
        _check(lib().orb_bf_query_packed_device(self.h, _dp(d_q), d_q.shape[0], _dp(d_packed), _stream(stream)))

    def query_packed_noalloc(self, d_q, d_packed, stream=None):
        """The same through the entry point that allocates nothing (fixed partials block in the library)."""
        _check(lib().orb_bf_query_packed_noalloc(self.h, _dp(d_q), d_q.shape[0], _dp(d_packed), _stream(stream)))

    def clock_probe(self, enable=True):
        """Engine clock (MHz) the scans since the previous call ran at (0.0 if none ran with the probe on); switches
        the probe on or off for the scans that follow.  Waits for the device."""
        mhz = C.c_double(0.0)
        _check(lib().orb_bf_clock_probe(self.h, int(enable), C.byref(mhz)))
        return mhz.value


class MultiBruteForce:
    """Brute-force top-2 over a database row-sharded over several devices of this process (RCCL all-gather)."""

    def __init__(self, devices):
        self.h = C.c_void_p()
        self._destroy = lib().orb_bf_multi_destroy
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().orb_bf_multi_create(devs, len(devices), C.byref(self.h)))
        self.exchange = "rccl" if lib().orb_bf_multi_exchange(self.h) == 1 else "copy"

    def close(self):
        if getattr(self, "h", None) and self.h:
            self._destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def set_database(self, db):
        db = np.ascontiguousarray(db, np.uint8)
        _check(lib().orb_bf_multi_set_database(self.h, _p(db), C.c_int64(len(db))))

    def query_top2(self, q):
        q = np.ascontiguousarray(q, np.uint8)
        nq = len(q)
        idx = np.zeros(nq, np.int32)
        d1 = np.zeros(nq, np.uint16)
        d2 = np.zeros(nq, np.uint16)
        _check(lib().orb_bf_multi_query_top2(self.h, _p(q), nq, _p(idx), _p(d1), _p(d2)))
        return idx, d1, d2


def extract_batch_multi(contexts, imgs):
    """orb_extract_batch_multi: contiguous frame blocks over several contexts (one per device)."""
    imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
    n = len(imgs)
    h, w = imgs[0].shape
    cap = contexts[0].cap
    ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
    hs = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    kps = np.zeros((n, cap), KP_DTYPE)
    desc = np.zeros((n, cap, 32), np.uint8)
    cnt = np.zeros(n, np.int32)
    _check(lib().orb_extract_batch_multi(hs, len(contexts), ptrs, n, w, h, imgs[0].strides[0], _p(kps), _p(desc), cap, _p(cnt)))
    return [(kps[f, : cnt[f]].copy(), desc[f, : cnt[f]].copy()) for f in range(n)]


def extract_batch_multi_device(contexts, d_blocks, d_kps, d_desc, d_counts, streams=None):
    """orb_extract_batch_multi_device: block r = d_blocks[r] (uint8 CUDA tensor [n_r, h, w] on contexts[r]'s device) with
    outputs d_kps[r] [n_r, cap, 28], d_desc[r] [n_r, cap, 32], d_counts[r] [n_r].  The block sizes must be the
    contiguous split of their total over the contexts.  streams: torch streams, one per context (asynchronous), or None
    (the contexts' own streams; returns when all blocks are done)."""
    R = len(contexts)
    n = sum(int(b.shape[0]) for b in d_blocks)
    assert [int(b.shape[0]) for b in d_blocks] == [(n * (r + 1)) // R - (n * r) // R for r in range(R)]
    ref = next(b for b in d_blocks if b.shape[0] > 0)
    h, w = int(ref.shape[1]), int(ref.shape[2])
    assert all(b.shape[0] == 0 or (b.shape[1:] == ref.shape[1:] and b.stride() == ref.stride()) for b in d_blocks)
    cap = int(d_kps[0].shape[1])
    arr = lambda ts: (C.c_void_p * R)(*[t.data_ptr() if t.shape[0] > 0 else None for t in ts])
    hs = (C.c_void_p * R)(*[c.h for c in contexts])
    st = (C.c_void_p * R)(*[s.cuda_stream for s in streams]) if streams is not None else None
    _check(lib().orb_extract_batch_multi_device(hs, R, arr(d_blocks), n, w, h, ref.stride(1), C.c_int64(ref.stride(0)), arr(d_kps),
                                                arr(d_desc), cap, arr(d_counts), st))


def alloc_host(shape, dtype=np.uint8):
    """A numpy array over page-locked host memory (orb_alloc_host): frames kept in it reach the device without a
    staging copy.  The memory is released when the array (and every view of it) is gone."""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _check(lib().orb_alloc_host(C.c_size_t(max(nbytes, 1)), C.byref(p)))
    free = lib().orb_free_host
    addr = p.value

    class _Owner:
        def __del__(self):
            free(C.c_void_p(addr))

    buf = (C.c_ubyte * max(nbytes, 1)).from_address(addr)
    buf._owner = _Owner()  # the ctypes block keeps the owner alive; numpy keeps the ctypes block as its base
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def merge_packed_device(device, d_gathered, nshards, nq, d_idx, d_best, d_second, stream=None):
    _check(lib().orb_bf_merge_packed_device(device, _dp(d_gathered), nshards, nq, _dp(d_idx), _dp(d_best), _dp(d_second), _stream(stream)))


class Vocabulary:
    """ORBVocabulary = TemplatedVocabulary<FORB::TDescriptor, FORB> with the tree on one device."""

    def __init__(self, handle):
        self.h = handle
        self._destroy = lib().orb_voc_destroy
        info = (C.c_int * 6)()
        _check(lib().orb_voc_info(self.h, info))
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = (int(v) for v in info)

    @classmethod
    def load(cls, path, device=0):
        """loadFromTextFile (the format of Vocabulary/ORBvoc.txt)."""
        h = C.c_void_p()
        _check(lib().orb_voc_load_text(str(path).encode(), device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, k, L, scoring, weighting, parent, leaf, desc, weight, device=0):
        parent = np.ascontiguousarray(parent, np.int32)
        leaf = np.ascontiguousarray(leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        _check(lib().orb_voc_create(k, L, scoring, weighting, _p(parent), _p(leaf), _p(desc), _p(weight), len(parent), device,
                                    C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "h", None) and self.h:
            self._destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def transform_features(self, desc, levelsup=4):
        """Per descriptor: (word id, word weight, node id `levelsup` levels above the leaves)."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        word = np.zeros(n, np.int32)
        node = np.zeros(n, np.int32)
        weight = np.zeros(n, np.float64)
        _check(lib().orb_voc_transform(self.h, _p(desc), n, levelsup, _p(word), _p(weight), _p(node)))
        return word, weight, node

    def transform_features_device(self, d_desc, levelsup, d_word, d_weight, d_node, stream=None):
        """d_desc uint8 [n, 32]; outputs int32 [n], float64 [n], int32 [n] CUDA tensors (any may be None)."""
        _check(lib().orb_voc_transform_device(self.h, _dp(d_desc), d_desc.shape[0], levelsup, _dp(d_word), _dp(d_weight),
                                              _dp(d_node), _stream(stream)))

    def transform(self, desc, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup) -> ((ids, values), (nodes, offsets, indices))."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        ids = np.zeros(max(n, 1), np.int32)
        vals = np.zeros(max(n, 1), np.float64)
        nodes = np.zeros(max(n, 1), np.int32)
        offs = np.zeros(n + 1, np.int32)
        idx = np.zeros(max(n, 1), np.int32)
        nb, nn = C.c_int(0), C.c_int(0)
        _check(lib().orb_voc_bow(self.h, _p(desc), n, levelsup, _p(ids), _p(vals), C.byref(nb), _p(nodes), _p(offs), _p(idx),
                                 C.byref(nn)))
        return (ids[: nb.value].copy(), vals[: nb.value].copy()), (
            nodes[: nn.value].copy(), offs[: nn.value + 1].copy(), idx[: offs[nn.value]].copy())


def bow_score_l1(bow1, bow2):
    """L1Scoring::score of two normalised bag-of-words vectors given as (ids, values)."""
    i1, v1 = np.ascontiguousarray(bow1[0], np.int32), np.ascontiguousarray(bow1[1], np.float64)
    i2, v2 = np.ascontiguousarray(bow2[0], np.int32), np.ascontiguousarray(bow2[1], np.float64)
    lib().orb_bow_score_l1.restype = C.c_double
    return lib().orb_bow_score_l1(_p(i1), _p(v1), len(i1), _p(i2), _p(v2), len(i2))
