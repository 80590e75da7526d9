"""ctypes bindings for the two CPU checkers -- TEST INFRASTRUCTURE ONLY.

* ``Oracle("port")``      -> oracle/libmimc3_oracle.so   (this repo's C restatement, mimc3_oracle.c)
* ``Oracle("reference")`` -> oracle/_ref/libmimc3_ref.so (the unmodified reference sources compiled
  by oracle/Makefile together with ref_harness.c; exists only where it was built in the container
  that holds /root/reference -- it travels to the GPU box as a prebuilt, git-ignored file)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing under mimc3_amd/ imports it: the product path has no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {
    "port": os.path.join(_HERE, "libmimc3_oracle.so"),
    "reference": os.path.join(_HERE, "_ref", "libmimc3_ref.so"),
}
_PREFIX = {"port": "orc_", "reference": "ref_"}

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def available(kind):
    return os.path.exists(_PATHS[kind])


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        path = _PATHS[kind]
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing -- run `make -C oracle {'oracle' if kind == 'port' else 'ref'}`")
        self.lib = C.CDLL(path)
        p = _PREFIX[kind]
        self._piv = getattr(self.lib, p + "get_uv_pivot")
        self._piv.restype = C.c_int64
        self._piv.argtypes = [_f64p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                              C.c_int32, C.c_int32, _i64p, _i32p, C.c_int64]
        self._match = getattr(self.lib, p + ("match_ncc_dlc" if kind == "port" else "matching_ncc_dlc_2"))
        self._match.restype = C.c_int
        self._match.argtypes = [_f32p, _f32p, C.c_int32, C.c_int32, _f64p, C.c_int32, _i32p, _i32p, _i64p,
                                C.c_int32, _f32p, C.c_int32]
        self._ruv = getattr(self.lib, p + "get_ruv_neighbor")
        self._ruv.restype = C.c_int32
        self._ruv.argtypes = [_f64p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _i32p, C.c_int32]
        if kind == "port":
            self._qm = self.lib.orc_qm_pseudosmooth
            self._qm.restype = C.c_int
            self._qm.argtypes = [C.c_int32, C.c_int32, _i32p, _f32p, _f32p, _i32p, C.c_int32, _f32p, C.c_int32,
                                 _i32p, _f64p, C.c_int32, _i64p]
        else:
            self._qm = self.lib.ref_get_dpf_pseudosmoothing
            self._qm.restype = C.c_int
            self._qm.argtypes = [C.c_int32, C.c_int32, _i32p, _f32p, _f32p, _i32p, C.c_int32, _f32p, C.c_int32,
                                 _i32p, _f64p]
            self._prep = self.lib.ref_postprocess_prep
            self._prep.restype = C.c_int32
            self._prep.argtypes = [_f32p, C.c_int32, _f64p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int32, _f32p, _i32p, _i32p, _f32p, _f32p]
        # N1: clustering, dpf0, dpf1
        self._clu = getattr(self.lib, p + "cluster_candidates")
        self._clu.restype = C.c_int32
        self._clu.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, _f32p, _i32p]
        self._d0 = getattr(self.lib, p + "get_dpf0")
        self._d0.restype = None
        if kind == "port":
            self._d0.argtypes = [_f32p, _i32p, C.c_int32, C.c_int32, C.c_float, _i32p]
        else:
            self._d0.argtypes = [_f32p, _i32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, _i32p]
        self._d1 = getattr(self.lib, p + "get_dpf1")
        self._d1.restype = C.c_int32
        self._d1.argtypes = [C.c_int32, C.c_int32, _i32p, _f32p, _f32p, _i32p, C.c_int32, _f32p, C.c_int32, _i32p, _f64p,
                             C.c_float, C.c_float]
        # N2: pre-filter
        self._conv = getattr(self.lib, p + "float_conv2")
        self._conv.restype = None
        self._conv.argtypes = [_f32p, C.c_int32, C.c_int32, _f32p, C.c_int32, C.c_int32, _f32p]
        # N4: control-point offset
        self._cp = getattr(self.lib, p + "get_offset_image")
        self._cp.restype = C.c_int
        base = [_f32p, _f32p, C.c_int32, C.c_int32, _f64p, C.c_int32, _i32p, C.c_float, C.c_int32, C.c_int32, C.c_float,
                C.c_float, C.POINTER(C.c_void_p), _i32p, C.c_uint32, _i32p, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]
        self._cp.argtypes = base + ([_i32p, _f32p] if kind == "port" else [])
        self._nthr = getattr(self.lib, p + "num_threads")
        self._nthr.restype = C.c_int

    def num_threads(self):
        return int(self._nthr())

    # -- a2 ---------------------------------------------------------------------------------
    def get_uv_pivot(self, xyuvav, dt, mpp, ocw, H, W, aw_sf=1.8, aw_cre=10.0, cap_per_point=256):
        xy = np.ascontiguousarray(xyuvav, np.float64)
        n = xy.shape[0]
        off = np.zeros(n + 1, np.int64)
        cap = int(cap_per_point) * n
        uv = np.zeros((cap, 2), np.int32)
        tot = self._piv(xy, n, dt, mpp, aw_sf, aw_cre, ocw, H, W, off, uv, cap)
        if tot < 0:
            raise ValueError(f"get_uv_pivot failed rc={tot} (-1 capacity, -2 a point with zero pivots)")
        return off, np.ascontiguousarray(uv[:tot])

    # -- a3..a7 -----------------------------------------------------------------------------
    def match(self, i0, i1, xyuvav, offset, piv_off, piv_uv, ocw, nthreads=0):
        i0 = np.ascontiguousarray(i0, np.float32)
        i1 = np.ascontiguousarray(i1, np.float32)
        xy = np.ascontiguousarray(xyuvav, np.float64)
        H, W = i0.shape
        n = xy.shape[0]
        out = np.empty((n, 3), np.float32)
        rc = self._match(i0, i1, H, W, xy, n, np.asarray(offset, np.int32), np.ascontiguousarray(piv_uv, np.int32),
                         np.ascontiguousarray(piv_off, np.int64), ocw, out, nthreads)
        if rc != 0:
            raise ValueError(f"match rc={rc}")
        return out

    # -- a8 ---------------------------------------------------------------------------------
    def get_ruv_neighbor(self, xyuvav, dimx, dimy, meter_per_spacing, radius):
        xy = np.ascontiguousarray(xyuvav, np.float64)
        cap = 4096
        ruv = np.zeros((cap, 2), np.int32)
        nn = self._ruv(xy, xy.shape[0], dimx, dimy, meter_per_spacing, radius, ruv, cap)
        if nn < 0:
            raise ValueError("neighbour capacity")
        return np.ascontiguousarray(ruv[:nn])

    # -- a9/a10 -----------------------------------------------------------------------------
    def qm(self, dpf, dpf_dx, dpf_dy, ruv, mvn, nclus, xyuvav, max_sweeps=101):
        """Returns (dpf, dx, dy, stats) -- copies; inputs untouched. stats is None for the reference."""
        dimy, dimx = dpf.shape
        d = np.array(dpf, np.int32, order="C")
        x = np.array(dpf_dx, np.float32, order="C")
        y = np.array(dpf_dy, np.float32, order="C")
        ruv = np.ascontiguousarray(ruv, np.int32)
        mvn = np.ascontiguousarray(mvn, np.float32)
        nclus = np.ascontiguousarray(nclus, np.int32)
        xy = np.ascontiguousarray(xyuvav, np.float64)
        kmax = mvn.shape[1]
        if self.kind == "port":
            stats = np.zeros(3, np.int64)
            self._qm(dimy, dimx, d, x, y, ruv, ruv.shape[0], mvn, kmax, nclus, xy, max_sweeps, stats)
            return d, x, y, stats
        self._qm(dimy, dimx, d, x, y, ruv, ruv.shape[0], mvn, kmax, nclus, xy)
        return d, x, y, None

    # -- N1 ---------------------------------------------------------------------------------
    def cluster_candidates(self, dp, kmax=32):
        """calc_mean_var_num_dp_cluster (:994-1130): dp [ndp][N][3] -> (mvn [N][kmax][5], nclus [N])."""
        dp = np.ascontiguousarray(dp, np.float32)
        ndp, n, _ = dp.shape
        mvn = np.zeros((n, kmax, 5), np.float32)
        nclus = np.zeros(n, np.int32)
        rc = self._clu(dp, ndp, n, kmax, mvn, nclus)
        if rc < 0:
            raise ValueError("more clusters than kmax")
        return mvn, nclus

    def get_dpf0(self, mvn, nclus, dimx, dimy, min_ratio=0.6):
        mvn = np.ascontiguousarray(mvn, np.float32)
        nclus = np.ascontiguousarray(nclus, np.int32)
        dpf = np.zeros(dimx * dimy, np.int32)
        if self.kind == "port":
            self._d0(mvn, nclus, dimx * dimy, mvn.shape[1], min_ratio, dpf)
        else:
            self._d0(mvn, nclus, dimx, dimy, mvn.shape[1], min_ratio, dpf)
        return dpf.reshape(dimy, dimx)

    def get_dpf1(self, dpf0, ruv, mvn, nclus, xyuvav, dt, mpp):
        """get_dpf1 (:1330-1718) -> (dpf, dx, dy); inputs untouched."""
        dimy, dimx = dpf0.shape
        d = np.array(dpf0, np.int32, order="C")
        x = np.zeros((dimy, dimx), np.float32)
        y = np.zeros((dimy, dimx), np.float32)
        ruv = np.ascontiguousarray(ruv, np.int32)
        mvn = np.ascontiguousarray(mvn, np.float32)
        self._d1(dimy, dimx, d.reshape(-1), x.reshape(-1), y.reshape(-1), ruv, ruv.shape[0], mvn, mvn.shape[1],
                 np.ascontiguousarray(nclus, np.int32), np.ascontiguousarray(xyuvav, np.float64), dt, mpp)
        return d, x, y

    # -- N2 ---------------------------------------------------------------------------------
    def float_conv2(self, img, kernel, out=None):
        """GMA_float_conv2 (:2517-2585). `out` is in/out (its border takes part); default zeros (T4)."""
        img = np.ascontiguousarray(img, np.float32)
        kernel = np.ascontiguousarray(kernel, np.float32)
        o = np.zeros_like(img) if out is None else np.array(out, np.float32, order="C")
        self._conv(img, img.shape[0], img.shape[1], kernel, kernel.shape[0], kernel.shape[1], o)
        return o

    # -- N4 ---------------------------------------------------------------------------------
    def get_offset_image(self, i0, i1, xyuvav, kernels, seed, vec_ocw=(7, 15, 30, 40), aw_cre=10.0, num_cp_max=500,
                         num_cp_min=50, ratio_cp=0.03, thres_spd_cp=10.0):
        """get_offset_image (:33-492) with the shuffle seed pinned. Returns (rc, offset[2], flag_cp[N], info, sduv);
        info/sduv are None for the reference."""
        i0 = np.ascontiguousarray(i0, np.float32); i1 = np.ascontiguousarray(i1, np.float32)
        xy = np.ascontiguousarray(xyuvav, np.float64)
        ks = [np.ascontiguousarray(k, np.float32) for k in kernels]
        kptr = (C.c_void_p * 3)(*[k.ctypes.data for k in ks])
        kdim = np.array([d for k in ks for d in k.shape], np.int32)
        off = np.zeros(2, np.int32)
        flag = np.zeros(xy.shape[0], np.uint8)
        args = [i0, i1, i0.shape[0], i0.shape[1], xy, xy.shape[0], np.array(vec_ocw, np.int32), aw_cre, num_cp_max, num_cp_min,
                ratio_cp, thres_spd_cp, kptr, kdim, seed, off, flag]
        if self.kind == "port":
            info = np.zeros(4, np.int32); sduv = np.zeros(2, np.float32)
            rc = self._cp(*args, info, sduv)
            return rc, off, flag, info, sduv
        rc = self._cp(*args)
        return rc, off, flag, None, None

    # -- reference only: candidates -> QM input (N1 rows, used to make realistic fixtures) ----
    def postprocess_prep(self, dp, xyuvav, dimx, dimy, dt, mpp, meter_per_spacing, radius_dpf1=3.0, kmax=32):
        assert self.kind == "reference"
        dp = np.ascontiguousarray(dp, np.float32)            # [ndp][N][3]
        ndp, n, _ = dp.shape
        xy = np.ascontiguousarray(xyuvav, np.float64)
        mvn = np.zeros((n, kmax, 5), np.float32)
        nclus = np.zeros(n, np.int32)
        dpf = np.zeros((dimy, dimx), np.int32)
        dx = np.zeros((dimy, dimx), np.float32)
        dy = np.zeros((dimy, dimx), np.float32)
        rc = self._prep(dp, ndp, xy, n, dimx, dimy, dt, mpp, meter_per_spacing, radius_dpf1, kmax, mvn, nclus,
                        dpf.reshape(-1), dx.reshape(-1), dy.reshape(-1))
        if rc < 0:
            raise ValueError("more clusters than kmax")
        return mvn, nclus, dpf, dx, dy
