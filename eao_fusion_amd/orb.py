"""ORBextractor -- same constructor arguments, call semantics and getters as the reference class
(reference include/ORBextractor.h:45-111); compute is eao_orb_* in libeaofusion_hip.so."""
import ctypes as C

import numpy as np

from . import _lib

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])  # cv::KeyPoint, 28 bytes

EDGE_THRESHOLD = 19


class ORBextractor:
    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST):
        self._L = _lib.load()
        self.nfeatures, self.scaleFactor, self.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        cfg = _lib.OrbCfg(self.nfeatures, self.scaleFactor, self.nlevels, self.iniThFAST, self.minThFAST)
        self._h = C.c_void_p()
        _lib.check(self._L.eao_orb_create(C.byref(cfg), C.byref(self._h)))
        n = self.nlevels
        self._scale, self._inv, self._s2, self._is2 = (np.zeros(n, np.float32) for _ in range(4))
        self._quota = np.zeros(n, np.int32)
        _lib.check(self._L.eao_orb_tables(self._h, _lib.ptr(self._scale), _lib.ptr(self._inv), _lib.ptr(self._s2),
                                         _lib.ptr(self._is2), _lib.ptr(self._quota)))
        self._last_shape = None

    def __del__(self):
        try:
            if self._h:
                self._L.eao_orb_destroy(self._h)
        except Exception:
            pass

    # --- reference getters (include/ORBextractor.h:63-83)
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self._scale.copy()

    def GetInverseScaleFactors(self):
        return self._inv.copy()

    def GetScaleSigmaSquares(self):
        return self._s2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self._is2.copy()

    @property
    def mnFeaturesPerLevel(self):
        return self._quota.copy()

    def max_keypoints(self, width, height):
        cap = C.c_int32()
        _lib.check(self._L.eao_orb_max_keypoints(self._h, width, height, C.byref(cap)))
        return cap.value

    # --- operator()(image, mask, keypoints, descriptors): mask is ignored upstream too (include/ORBextractor.h:58)
    def __call__(self, image, mask=None):
        if image is None or image.size == 0:
            return None, None  # reference returns without touching its outputs (src/ORBextractor.cc:1046-1047)
        kps, desc = self.extract_batch(image[None])
        return kps[0], desc[0]

    def extract_batch(self, images):
        """images: (B, H, W) uint8.  Returns per-frame lists of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images)
        assert images.dtype == np.uint8 and images.ndim == 3, "CV_8UC1 images expected (reference asserts the same)"
        B, H, W = images.shape
        cap = self.max_keypoints(W, H)
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        _lib.check(self._L.eao_orb_extract_batch(self._h, _lib.ptr(images), W, H, W, H * W, B, _lib.ptr(kps), _lib.ptr(desc),
                                                cap, _lib.ptr(n)))
        self._last_shape = (B, H, W)
        return [kps[f, :n[f]] for f in range(B)], [desc[f, :n[f]] for f in range(B)]

    def extract_batch_device(self, d_img, width, height, stride, frame_stride, batch, d_kps, d_desc, cap, d_n, stream=0):
        """Device-resident variant: all pointers are integers (HBM addresses), asynchronous on `stream`."""
        _lib.check(self._L.eao_orb_extract_batch_device(self._h, d_img, width, height, stride, frame_stride, batch, d_kps,
                                                       d_desc, cap, d_n, stream))
        self._last_shape = (batch, height, width)

    # --- streaming host API (eao_orb_stream_*): pinned slots, asynchronous submit, upload / extraction / download overlapped
    def stream_create(self, width, height, batch, nslots=3):
        _lib.check(self._L.eao_orb_stream_create(self._h, width, height, batch, nslots))
        self._stream = (width, height, batch, nslots)
        self._slots = []
        for s in range(nslots):
            sl = _lib.OrbSlot()
            _lib.check(self._L.eao_orb_stream_slot(self._h, s, C.byref(sl)))
            frames = np.ctypeslib.as_array(C.cast(sl.frames, C.POINTER(C.c_uint8)), shape=(batch, height, sl.stride))
            kps = np.ctypeslib.as_array(C.cast(sl.kps, C.POINTER(C.c_uint8)), shape=(batch, sl.cap, 28)).view(KP_DTYPE).reshape(batch, sl.cap)
            desc = np.ctypeslib.as_array(C.cast(sl.desc, C.POINTER(C.c_uint8)), shape=(batch, sl.cap, 32))
            n = np.ctypeslib.as_array(C.cast(sl.n, C.POINTER(C.c_int32)), shape=(batch,))
            self._slots.append(dict(frames=frames[:, :, :width], frames_pitched=frames, kps=kps, desc=desc, n=n, cap=int(sl.cap)))
        return self._slots

    def stream_submit(self, slot, batch=None):
        _lib.check(self._L.eao_orb_stream_submit(self._h, slot, self._stream[2] if batch is None else batch))

    def stream_wait(self, slot):
        """Blocks until the slot's results are in its pinned arrays; returns per-frame (keypoints, descriptors) VIEWS into them."""
        _lib.check(self._L.eao_orb_stream_wait(self._h, slot))
        sl = self._slots[slot]
        B = len(sl["n"])
        return [sl["kps"][f, :sl["n"][f]] for f in range(B)], [sl["desc"][f, :sl["n"][f]] for f in range(B)]

    def set_profiling(self, on):
        _lib.check(self._L.eao_orb_set_profiling(self._h, 1 if on else 0))

    def last_timing(self):
        ms = (C.c_float * 6)()
        _lib.check(self._L.eao_orb_last_timing(self._h, ms))
        return dict(zip(("pyramid", "fast", "quadtree", "blur", "orient_describe", "total"), list(ms)))

    # --- stage taps
    def level_image(self, level, frame=0, blurred=False):
        w, h = C.c_int32(), C.c_int32()
        _lib.check(self._L.eao_orb_level(self._h, frame, level, 1 if blurred else 0, C.byref(w), C.byref(h), None))
        out = np.zeros((h.value, w.value), np.uint8)
        _lib.check(self._L.eao_orb_level(self._h, frame, level, 1 if blurred else 0, None, None, _lib.ptr(out)))
        return out

    def level_candidates(self, level, frame=0):
        n = C.c_int32()
        _lib.check(self._L.eao_orb_level_candidates(self._h, frame, level, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3), np.float32)
        if n.value:
            _lib.check(self._L.eao_orb_level_candidates(self._h, frame, level, _lib.ptr(out), n.value, C.byref(n)))
        return out

    @property
    def mvImagePyramid(self):
        """Level images of frame 0 of the last call with the reference's 19 px BORDER_REFLECT_101 frame around them
        (src/ORBextractor.cc:1113-1128), through eao_orb_pyramid: one launch into the handle's pinned block (the hot path itself
        never reads the border).  Returned as copies of the bordered images."""
        return self.image_pyramid(EDGE_THRESHOLD)

    def image_pyramid(self, border, frame=0):
        lv = (_lib.OrbLevelView * 16)()
        _lib.check(self._L.eao_orb_pyramid(self._h, int(frame), int(border), lv))
        out = []
        for l in range(self.nlevels):
            v = lv[l]
            rows, cols = v.height + 2 * border, v.width + 2 * border
            base = v.data - border * v.step - border
            buf = (C.c_uint8 * (rows * v.step)).from_address(base)
            out.append(np.frombuffer(buf, np.uint8).reshape(rows, v.step)[:, :cols].copy())
        return out

    def set_keep_pyramid(self, border):
        _lib.check(self._L.eao_orb_set_keep_pyramid(self._h, int(border)))

    def extract_ref(self, img):
        """eao_orb_extract_ref: the class-surface adapter's call (results read in place from the handle's pinned block, copied here)."""
        img = np.ascontiguousarray(img, np.uint8)
        kp, dp, n = C.c_void_p(), C.c_void_p(), C.c_int32()
        _lib.check(self._L.eao_orb_extract_ref(self._h, _lib.ptr(img), img.shape[1], img.shape[0], img.strides[0], C.byref(kp), C.byref(dp), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        k = np.frombuffer((C.c_uint8 * (n.value * 28)).from_address(kp.value), KP_DTYPE).copy()
        d = np.frombuffer((C.c_uint8 * (n.value * 32)).from_address(dp.value), np.uint8).reshape(n.value, 32).copy()
        return k, d


def compute_stereo_matches(ext_left, ext_right, kps_l, desc_l, kps_r, desc_r, mb, mbf, frame=0):
    """Frame::ComputeStereoMatches (reference src/Frame.cc:841-1013) on the pyramids the two extractors still hold on the
    device after extracting the stereo pair.  kps_*: KP_DTYPE arrays (mvKeys / mvKeysRight), desc_*: (n, 32) uint8.
    Returns (u_right, depth), float32 arrays with -1 where there is no match."""
    kl = np.ascontiguousarray(kps_l, KP_DTYPE)
    kr = np.ascontiguousarray(kps_r, KP_DTYPE)
    dl = np.ascontiguousarray(desc_l, np.uint8)
    dr = np.ascontiguousarray(desc_r, np.uint8)
    ur = np.full(len(kl), -1, np.float32)
    dp = np.full(len(kl), -1, np.float32)
    _lib.check(_lib.load().eao_compute_stereo_matches(ext_left._h, ext_right._h, int(frame), len(kl), _lib.ptr(kl), _lib.ptr(dl),
                                                      len(kr), _lib.ptr(kr), _lib.ptr(dr), float(mb), float(mbf), _lib.ptr(ur), _lib.ptr(dp)))
    return ur, dp
