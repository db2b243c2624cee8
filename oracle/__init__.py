"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED, with one exception: the
bag-of-words bookkeeping (BowVector / FeatureVector) is checked against the reference's own two
source files compiled here (oracle/ref_dbow2/Makefile -> oracle/_ref/libdbow2_ref.so).

ctypes binding over oracle/liborb_oracle.so (built by oracle/Makefile from
the C++ restatement in this directory).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liborb_oracle.so")

Q_RESIZE_AREA, Q_UMAX_FORK, Q_DIVIDE_N3, Q_GRID_EXCLUSIVE, Q_WINDOW_CIRCULAR, Q_GAUSS_LEGACY, Q_LIBM_TRIG = (
    1 << i for i in range(7)
)
QUIRKS_FORK = Q_RESIZE_AREA | Q_UMAX_FORK | Q_DIVIDE_N3 | Q_GRID_EXCLUSIVE | Q_WINDOW_CIRCULAR
QUIRKS_CANONICAL = 0

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
assert KP_DTYPE.itemsize == 28


def build(force=False):
    """Compile the oracle with its Makefile (g++ only; no reference build exists)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oro_create.restype = C.c_void_p
        _lib.oro_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint]
        _lib.oro_destroy.argtypes = [C.c_void_p]
        _lib.oro_fast_atan2.restype = C.c_float
        _lib.oro_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.oro_ic_angle.restype = C.c_float
        _lib.oro_sincos_mismatch.restype = C.c_long
        _lib.oro_sincos_mismatch.argtypes = [C.c_float, C.c_float, C.c_int, C.POINTER(C.c_long)]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class Extractor:
    """ORBextractor restatement (src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, quirks=QUIRKS_FORK):
        self.nlevels, self.quirks = nlevels, quirks
        self.h = C.c_void_p(lib().oro_create(nfeatures, C.c_float(scale), nlevels, ini_th, min_th, quirks))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oro_destroy(self.h)
            self.h = None

    def extract(self, img, cap=None):
        img = _u8(img)
        hgt, w = img.shape
        cap = cap or 65536
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = lib().oro_extract(self.h, _p(img), w, hgt, img.strides[0], _p(kps), _p(desc), cap)
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def tables(self):
        L = self.nlevels
        f = [np.zeros(L, dtype=np.float32) for _ in range(4)]
        per = np.zeros(L, dtype=np.int32)
        um = np.zeros(16, dtype=np.int32)
        lib().oro_get_tables(self.h, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(per), _p(um))
        return dict(scale=f[0], inv_scale=f[1], sigma2=f[2], inv_sigma2=f[3], per_level=per, umax=um)

    def compute_pyramid(self, img):
        img = _u8(img)
        lib().oro_compute_pyramid(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0])

    def level(self, level, blurred=False):
        w, hgt = C.c_int(), C.c_int()
        assert lib().oro_level_size(self.h, level, C.byref(w), C.byref(hgt)) == 0
        out = np.zeros((hgt.value, w.value), dtype=np.uint8)
        if lib().oro_get_level(self.h, 1 if blurred else 0, level, _p(out)) != 0:
            return None
        return out

    def level_keys(self, level, after_quadtree=True):
        which = 1 if after_quadtree else 0
        n = lib().oro_get_level_keys(self.h, which, level, None, 0)
        out = np.zeros(max(n, 1), dtype=KP_DTYPE)
        lib().oro_get_level_keys(self.h, which, level, _p(out), n)
        return out[:n]

    def distribute(self, keys, min_x, max_x, min_y, max_y, N):
        keys = np.ascontiguousarray(keys, dtype=KP_DTYPE)
        out = np.zeros(len(keys) + 8, dtype=KP_DTYPE)
        n = lib().oro_distribute(self.h, _p(keys), len(keys), min_x, max_x, min_y, max_y, N, _p(out), len(out))
        return out[:n].copy()


def resize(src, dw, dh, area=True):
    src = _u8(src)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    lib().oro_resize(1 if area else 0, _p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian(src, legacy=False):
    src = _u8(src)
    dst = np.zeros_like(src)
    lib().oro_gaussian(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0], int(legacy))
    return dst


def fast(img, threshold, nms=True):
    img = _u8(img)
    cap = img.size
    out = np.zeros(cap, dtype=KP_DTYPE)
    n = lib().oro_fast(_p(img), img.strides[0], img.shape[1], img.shape[0], threshold, int(nms), _p(out), cap)
    return out[:n].copy()


def fast_score(img, x, y, threshold):
    img = _u8(img)
    p = img.ctypes.data + y * img.strides[0] + x
    return lib().oro_fast_score(C.c_void_p(p), img.strides[0], threshold)


def fast_atan2(y, x):
    return lib().oro_fast_atan2(C.c_float(y), C.c_float(x))


def sincos(x, libm=False):
    s, c = C.c_float(), C.c_float()
    (lib().oro_sincos_libm if libm else lib().oro_sincos)(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def sincos_mismatch(lo, hi, stride_ulps):
    tot = C.c_long()
    bad = lib().oro_sincos_mismatch(lo, hi, stride_ulps, C.byref(tot))
    return bad, tot.value


def ic_angle(img, x, y, umax):
    img = _u8(img)
    um = np.ascontiguousarray(umax, dtype=np.int32)
    return lib().oro_ic_angle(_p(img), img.shape[1], img.shape[0], C.c_float(x), C.c_float(y), _p(um))


def descriptor(blurred, x, y, angle, libm=False):
    blurred = _u8(blurred)
    d = np.zeros(32, dtype=np.uint8)
    lib().oro_descriptor(_p(blurred), blurred.shape[1], blurred.shape[0], C.c_float(x), C.c_float(y), C.c_float(angle), int(libm), _p(d))
    return d


def hamming(a, b, bithack=True):
    a, b = _u8(a), _u8(b)
    return (lib().oro_hamming_bithack if bithack else lib().oro_hamming_popcount)(_p(a), _p(b))


def three_maxima(counts):
    c = np.ascontiguousarray(counts, dtype=np.int32)
    ind = np.zeros(3, dtype=np.int32)
    lib().oro_three_maxima(_p(c), len(c), _p(ind))
    return tuple(int(v) for v in ind)


def features_in_area(kps, img_w, img_h, x, y, r, min_level=-1, max_level=-1, quirks=QUIRKS_FORK):
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    out = np.zeros(len(kps) + 1, dtype=np.int32)
    n = lib().oro_features_in_area(
        _p(kps), len(kps), img_w, img_h, C.c_float(x), C.c_float(y), C.c_float(r), min_level, max_level, quirks, _p(out), len(out)
    )
    return out[:n].copy()


def search_for_initialization(k1, d1, k2, d2, img_w, img_h, prev_xy, window=100, nnratio=0.9, check_ori=True, quirks=QUIRKS_FORK):
    """ORBmatcher::SearchForInitialization.  Returns (nmatches, matches12, prev_xy_out)."""
    k1 = np.ascontiguousarray(k1, dtype=KP_DTYPE)
    k2 = np.ascontiguousarray(k2, dtype=KP_DTYPE)
    d1, d2 = _u8(d1), _u8(d2)
    prev = np.ascontiguousarray(prev_xy, dtype=np.float32).copy()
    m12 = np.zeros(max(len(k1), 1), dtype=np.int32)
    n = lib().oro_search_for_initialization(
        _p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), img_w, img_h, _p(prev), _p(m12), window, C.c_float(nnratio), int(check_ori), quirks
    )
    return n, m12[: len(k1)], prev


def search_by_projection(u, v, radius, min_level, max_level, qangle, qdesc, qobs, k2, d2, img_w, img_h, occupied2=None, th_high=100, check_ori=True, quirks=QUIRKS_FORK):
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    u, v, radius, qangle = f32(u), f32(v), f32(radius), f32(qangle)
    min_level, max_level = i32(min_level), i32(max_level)
    qdesc, d2 = _u8(qdesc), _u8(d2)
    qobs = _u8(qobs) if qobs is not None else None
    occ = _u8(occupied2) if occupied2 is not None else None
    k2 = np.ascontiguousarray(k2, dtype=KP_DTYPE)
    match2 = np.zeros(max(len(k2), 1), dtype=np.int32)
    n = lib().oro_search_by_projection(
        _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(qangle), _p(qdesc), _p(qobs), len(u),
        _p(k2), _p(d2), len(k2), img_w, img_h, _p(occ), _p(match2), th_high, int(check_ori), quirks,
    )
    return n, match2[: len(k2)]


def search_local_map(u, v, radius, min_level, max_level, qdesc, qobs, k2, d2, img_w, img_h, occupied2=None, th_high=100, nnratio=0.8, quirks=QUIRKS_FORK):
    """SearchByProjection(Frame&, const vector<MapPoint*>&, th) in kernel-level form: (nmatches, match2)."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    u, v, radius = f32(u), f32(v), f32(radius)
    min_level, max_level = i32(min_level), i32(max_level)
    qdesc, d2 = _u8(qdesc), _u8(d2)
    qobs = _u8(qobs) if qobs is not None else None
    occ = _u8(occupied2) if occupied2 is not None else None
    k2 = np.ascontiguousarray(k2, dtype=KP_DTYPE)
    match2 = np.zeros(max(len(k2), 1), dtype=np.int32)
    n = lib().oro_search_local_map(
        _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(qdesc), _p(qobs), len(u),
        _p(k2), _p(d2), len(k2), img_w, img_h, _p(occ), _p(match2), th_high, C.c_float(nnratio), quirks,
    )
    return n, match2[: len(k2)]


def search_best(u, v, radius, min_level, max_level, qdesc, k2, d2, img_w, img_h, inv_level_sigma2=None, chi2_max=5.99, th=50, quirks=QUIRKS_FORK):
    """Candidate loop of Fuse / SearchBySim3: (best_idx, best_dist) per query."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    u, v, radius = f32(u), f32(v), f32(radius)
    min_level, max_level = i32(min_level), i32(max_level)
    qdesc, d2 = _u8(qdesc), _u8(d2)
    k2 = np.ascontiguousarray(k2, dtype=KP_DTYPE)
    sig = f32(inv_level_sigma2) if inv_level_sigma2 is not None else None
    idx = np.zeros(max(len(u), 1), dtype=np.int32)
    dist = np.zeros(max(len(u), 1), dtype=np.int32)
    lib().oro_search_best(
        _p(u), _p(v), _p(radius), _p(min_level), _p(max_level), _p(qdesc), len(u), _p(k2), _p(d2), len(k2), img_w, img_h,
        _p(sig), 0 if sig is None else len(sig), C.c_double(chi2_max), th, quirks, _p(idx), _p(dist),
    )
    return idx[: len(u)], dist[: len(u)]


def bf_top2(q, db):
    q, db = _u8(q), _u8(db)
    nq, nd = len(q), len(db)
    idx = np.zeros(nq, dtype=np.int32)
    d1 = np.zeros(nq, dtype=np.uint16)
    d2 = np.zeros(nq, dtype=np.uint16)
    lib().oro_bf_top2(_p(q), nq, _p(db), C.c_longlong(nd), _p(idx), _p(d1), _p(d2))
    return idx, d1, d2


def distinctive_descriptor(desc, self_distance=False):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:186-214) on an (n, 32) uint8 array:
    (index of the chosen descriptor, its median distance); (-1, None) for n == 0."""
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
    med = C.c_int(0)
    i = lib().oro_distinctive(_p(desc), len(desc), int(self_distance), C.byref(med))
    return (i, med.value) if i >= 0 else (-1, None)


def set_image_bounds(bounds):
    """Frame::mnMinX, mnMaxX, mnMinY, mnMaxY for the grid functions above (None = the image rectangle)."""
    if bounds is None:
        lib().oro_set_image_bounds(None)
    else:
        b = np.ascontiguousarray(bounds, dtype=np.float32)
        lib().oro_set_image_bounds(_p(b))


def undistort_keypoints(kps, K, dist):
    """Frame::UndistortKeyPoints; K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2, k3)."""
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    out = np.zeros(len(kps), dtype=KP_DTYPE)
    K = np.ascontiguousarray(K, dtype=np.float32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    lib().oro_undistort_keypoints(_p(kps), _p(out), len(kps), _p(K), _p(dist))
    return out


def undistort_points(xy, K, dist):
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    K = np.ascontiguousarray(K, dtype=np.float32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    lib().oro_undistort_points(_p(xy), _p(out), len(xy), _p(K), _p(dist))
    return out


def compute_image_bounds(cols, rows, K, dist):
    """Frame::ComputeImageBounds -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    K = np.ascontiguousarray(K, dtype=np.float32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    b = np.zeros(4, dtype=np.float32)
    lib().oro_compute_image_bounds(cols, rows, _p(K), _p(dist), _p(b))
    return b


# ---- DBoW2 vocabulary tree, bag of words, SearchByBoW ------------------------------------------------
class Vocabulary:
    """TemplatedVocabulary<FORB::TDescriptor, FORB> (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("vocabulary could not be loaded")
        self.h = C.c_void_p(handle)
        info = np.zeros(6, dtype=np.int32)
        lib().oro_voc_info(self.h, _p(info))
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = (int(v) for v in info)

    @classmethod
    def from_arrays(cls, k, L, scoring, weighting, parent, leaf, desc, weight):
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        leaf = _u8(leaf)
        desc = _u8(desc)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        lib().oro_voc_from_arrays.restype = C.c_void_p
        return cls(lib().oro_voc_from_arrays(k, L, scoring, weighting, _p(parent), _p(leaf), _p(desc), _p(weight), len(parent)))

    @classmethod
    def load(cls, path):
        lib().oro_voc_load.restype = C.c_void_p
        return cls(lib().oro_voc_load(str(path).encode()))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oro_voc_free(self.h)
            self.h = None

    def transform_features(self, desc, levelsup=4):
        """Per descriptor: (word id, word weight, node id `levelsup` levels above the leaves)."""
        desc = _u8(desc).reshape(-1, 32)
        n = len(desc)
        word = np.zeros(n, dtype=np.int32)
        node = np.zeros(n, dtype=np.int32)
        weight = np.zeros(n, dtype=np.float64)
        lib().oro_voc_transform_features(self.h, _p(desc), n, levelsup, _p(word), _p(weight), _p(node))
        return word, weight, node

    def transform(self, desc, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup) -> ((ids, values), (nodes, offsets, indices))."""
        desc = _u8(desc).reshape(-1, 32)
        n = len(desc)
        ids = np.zeros(max(n, 1), dtype=np.int32)
        vals = np.zeros(max(n, 1), dtype=np.float64)
        nodes = np.zeros(max(n, 1), dtype=np.int32)
        offs = np.zeros(n + 1, dtype=np.int32)
        idx = np.zeros(max(n, 1), dtype=np.int32)
        nb, nn = C.c_int(0), C.c_int(0)
        lib().oro_voc_bow(self.h, _p(desc), n, levelsup, _p(ids), _p(vals), C.byref(nb), _p(nodes), _p(offs), _p(idx), C.byref(nn))
        return (ids[: nb.value].copy(), vals[: nb.value].copy()), (
            nodes[: nn.value].copy(), offs[: nn.value + 1].copy(), idx[: offs[nn.value]].copy())


def score_l1(bow1, bow2):
    """L1Scoring::score of two normalised bag-of-words vectors given as (ids, values)."""
    i1, v1 = np.ascontiguousarray(bow1[0], np.int32), np.ascontiguousarray(bow1[1], np.float64)
    i2, v2 = np.ascontiguousarray(bow2[0], np.int32), np.ascontiguousarray(bow2[1], np.float64)
    lib().oro_score_l1.restype = C.c_double
    return lib().oro_score_l1(_p(i1), _p(v1), len(i1), _p(i2), _p(v2), len(i2))


def search_by_bow(kf_keys_un, kf_desc, kf_valid, kf_fv, f_keys, f_desc, f_fv, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...): returns (nmatches, matchKF[nF]); kf_fv / f_fv are
    feature vectors (nodes, offsets, indices) as Vocabulary.transform returns them."""
    kk = np.ascontiguousarray(kf_keys_un, dtype=KP_DTYPE)
    fk = np.ascontiguousarray(f_keys, dtype=KP_DTYPE)
    kd, fd, kv = _u8(kf_desc), _u8(f_desc), _u8(kf_valid)
    a = [np.ascontiguousarray(x, dtype=np.int32) for x in kf_fv]
    b = [np.ascontiguousarray(x, dtype=np.int32) for x in f_fv]
    match = np.full(max(len(fk), 1), -1, dtype=np.int32)
    n = lib().oro_search_by_bow(_p(kk), _p(kd), _p(kv), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]), _p(fk), _p(fd), len(fk),
                                _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]), C.c_float(nnratio), int(check_ori), _p(match))
    return n, match[: len(fk)]


def search_by_bow_keyframes(keys_un1, desc1, valid1, fv1, keys_un2, desc2, valid2, fv2, nnratio=0.75, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...): returns (nmatches, matches12[n1])."""
    k1 = np.ascontiguousarray(keys_un1, dtype=KP_DTYPE)
    k2 = np.ascontiguousarray(keys_un2, dtype=KP_DTYPE)
    d1, d2, v1, v2 = _u8(desc1), _u8(desc2), _u8(valid1), _u8(valid2)
    a = [np.ascontiguousarray(x, dtype=np.int32) for x in fv1]
    b = [np.ascontiguousarray(x, dtype=np.int32) for x in fv2]
    match = np.full(max(len(k1), 1), -1, dtype=np.int32)
    n = lib().oro_search_by_bow_keyframes(_p(k1), _p(d1), _p(v1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                          _p(k2), _p(d2), _p(v2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                          C.c_float(nnratio), int(check_ori), _p(match))
    return n, match[: len(k1)]


def search_for_triangulation(keys_un1, desc1, free1, fv1, keys_un2, desc2, free2, fv2, F12, epipole, scale_factors,
                             level_sigma2, mark_matched=False, check_ori=True):
    """ORBmatcher::SearchForTriangulation (monocular): returns (nmatches, matches12[n1])."""
    k1 = np.ascontiguousarray(keys_un1, dtype=KP_DTYPE)
    k2 = np.ascontiguousarray(keys_un2, dtype=KP_DTYPE)
    d1, d2, f1, f2 = _u8(desc1), _u8(desc2), _u8(free1), _u8(free2)
    a = [np.ascontiguousarray(x, dtype=np.int32) for x in fv1]
    b = [np.ascontiguousarray(x, dtype=np.int32) for x in fv2]
    F = np.ascontiguousarray(F12, dtype=np.float32).reshape(9)
    ep = None if epipole is None else np.ascontiguousarray(epipole, dtype=np.float32)
    sc = np.ascontiguousarray(scale_factors, dtype=np.float32)
    sg = np.ascontiguousarray(level_sigma2, dtype=np.float32)
    match = np.full(max(len(k1), 1), -1, dtype=np.int32)
    n = lib().oro_search_for_triangulation(_p(k1), _p(d1), _p(f1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                           _p(k2), _p(d2), _p(f2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                           _p(F), _p(ep), _p(sc), _p(sg), len(sc), int(mark_matched), int(check_ori), _p(match))
    return n, match[: len(k1)]


def _bow_stream(fn, ids, weights, nids, scoring, weighting):
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    nids = np.ascontiguousarray(nids, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    n = len(ids)
    bi = np.zeros(max(n, 1), dtype=np.int32)
    bv = np.zeros(max(n, 1), dtype=np.float64)
    nodes = np.zeros(max(n, 1), dtype=np.int32)
    offs = np.zeros(n + 1, dtype=np.int32)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    nb, nn = C.c_int(0), C.c_int(0)
    fn(_p(ids), _p(weights), _p(nids), n, int(scoring), int(weighting), _p(bi), _p(bv), C.byref(nb), _p(nodes), _p(offs),
       _p(idx), C.byref(nn))
    return (bi[: nb.value].copy(), bv[: nb.value].copy()), (
        nodes[: nn.value].copy(), offs[: nn.value + 1].copy(), idx[: offs[nn.value]].copy())


def bow_accumulate(ids, weights, nids, scoring, weighting):
    """The bookkeeping half of TemplatedVocabulary::transform on a stream of per-feature (word id, weight, node id):
    ((ids, values), (nodes, offsets, indices)), the oracle's restatement."""
    return _bow_stream(lib().oro_bow_accumulate, ids, weights, nids, scoring, weighting)


_REF_DBOW2 = None


def ref_dbow2_available():
    return os.path.exists(os.path.join(_DIR, "_ref", "libdbow2_ref.so"))


def ref_bow_accumulate(ids, weights, nids, scoring, weighting):
    """The same through the reference's own BowVector.cpp / FeatureVector.cpp, compiled by oracle/ref_dbow2/Makefile."""
    global _REF_DBOW2
    if _REF_DBOW2 is None:
        _REF_DBOW2 = C.CDLL(os.path.join(_DIR, "_ref", "libdbow2_ref.so"))
    return _bow_stream(_REF_DBOW2.ref_bow_accumulate, ids, weights, nids, scoring, weighting)
