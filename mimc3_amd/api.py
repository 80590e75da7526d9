"""Host-side mirror of the reference's entry points for the hot path, over the C ABI
(include/mimc3_hip.h -> csrc/libmimc3_hip.so).

Names and argument meaning follow the reference (MIMC_module.h:41-58):
    get_uv_pivot, matching_ncc_dlc_2, get_ruv_neighbor, get_dpf_pseudosmoothing
with the ragged ``GMA_int32 **uv_pivot`` flattened to CSR ``(piv_off, piv_uv)`` and the ragged
``GMA_float **mvn_dp`` padded to ``[N][Kmax][5]`` + ``nclus[N]``.

Importing this module loads the HIP library and raises if it is missing: there is no CPU fallback
(the CPU restatement lives in oracle/ and is test infrastructure only).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIMC3_HIP_LIB") or os.path.join(_HERE, "csrc", "libmimc3_hip.so")   # (override: A/B builds in tools/)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not built. Run `make -C mimc3_amd/csrc` (or python -c 'import __graft_entry__ as g; g.build()'). "
        "mimc3_amd has no CPU fallback.")
_lib = C.CDLL(LIB_PATH)

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_vp = C.c_void_p

# every symbol include/mimc3_hip.h declares (checked by tests/test_capi_symbols.py)
_lib.mimc3_last_error.restype = C.c_char_p
_lib.mimc3_version.restype = C.c_char_p
_lib.mimc3_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
_lib.mimc3_ctx_destroy.argtypes = [_vp]
_lib.mimc3_ctx_destroy.restype = None
_lib.mimc3_ctx_set_images.argtypes = [_vp, _f32p, _f32p, C.c_int32, C.c_int32]
_lib.mimc3_ctx_set_images_dev.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32]
_lib.mimc3_ctx_set_images_u8.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32]
_lib.mimc3_ctx_set_images_u16.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32]
_lib.mimc3_host_alloc.argtypes = [C.c_size_t]
_lib.mimc3_host_alloc.restype = _vp
_lib.mimc3_host_free.argtypes = [_vp]
_lib.mimc3_host_free.restype = None
_lib.mimc3_get_uv_pivot.argtypes = [_f64p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                    C.c_int32, _i64p, _vp, C.c_int64, C.POINTER(C.c_int64)]
_lib.mimc3_match_ncc_dlc.argtypes = [_vp, _f64p, C.c_int32, _i32p, _i32p, _i64p, C.c_int32, C.c_int32, _f32p]
_lib.mimc3_match_ncc_dlc_dev.argtypes = [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, _vp, _vp]
_lib.mimc3_pivot_corridors.argtypes = [_f64p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _vp]
_lib.mimc3_get_uv_pivot_dev.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int64, C.POINTER(C.c_int64), _i32p, _vp]
_lib.mimc3_match_ncc_dlc_geo.argtypes = [_vp, _f64p, C.c_int32, _i32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _f32p]
_lib.mimc3_match_ncc_dlc_cor.argtypes = [_vp, _f64p, _vp, C.c_int32, _i32p, C.c_int32, C.c_int32, _f32p]
_lib.mimc3_qm_launches_per_sweep.restype = C.c_int32
_lib.mimc3_pivot_extent.argtypes = [_i32p, _i64p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32)]
_lib.mimc3_get_ruv_neighbor.argtypes = [_f64p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _i32p, C.c_int32,
                                        C.POINTER(C.c_int32)]
_lib.mimc3_qm_pseudosmooth.argtypes = [_vp, C.c_int32, C.c_int32, _i32p, _f32p, _f32p, _i32p, C.c_int32, _f32p,
                                       C.c_int32, _i32p, _f64p, C.c_int32, C.POINTER(C.c_int32)]
_lib.mimc3_qm_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
_lib.mimc3_qm_workspace_bytes.restype = C.c_int64
_lib.mimc3_qm_pseudosmooth_dev.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp,
                                           _vp, C.c_int32, _vp, _vp, _vp]
_lib.mimc3_cluster_candidates.argtypes = [_vp, _f32p, C.c_int32, C.c_int32, C.c_int32, _f32p, _i32p, C.POINTER(C.c_int32)]
_lib.mimc3_cluster_candidates_dev.argtypes = [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]
_lib.mimc3_get_dpf0.argtypes = [_vp, _f32p, _i32p, C.c_int32, C.c_int32, C.c_float, _i32p]
_lib.mimc3_get_dpf0_dev.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_float, _vp, _vp]
_lib.mimc3_get_dpf1.argtypes = [_vp, C.c_int32, C.c_int32, _i32p, _f32p, _f32p, _i32p, C.c_int32, _f32p, C.c_int32, _i32p,
                                _f64p, C.c_float, C.c_float, C.POINTER(C.c_int32)]
_lib.mimc3_dpf1_workspace_bytes.argtypes = [C.c_int32]
_lib.mimc3_dpf1_workspace_bytes.restype = C.c_int64
_lib.mimc3_get_dpf1_dev.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp,
                                    C.c_float, C.c_float, _vp, C.POINTER(C.c_int32), _vp]
_lib.mimc3_float_conv2.argtypes = [_vp, _f32p, C.c_int32, C.c_int32, _f32p, C.c_int32, C.c_int32, _f32p]
_lib.mimc3_float_conv2_dev.argtypes = [_vp, _vp, C.c_int32, C.c_int32, _f32p, C.c_int32, C.c_int32, _vp, _vp, _vp]
_lib.mimc3_ctx_filter_images.argtypes = [_vp, _vp, C.c_int32, C.c_int32]
_lib.mimc3_ctx_get_images.argtypes = [_vp, _vp, _vp]


class CpParams(C.Structure):
    """mimc3_cp_params (include/mimc3_hip.h): the reference's globals that get_offset_image reads."""
    _fields_ = [("vec_ocw", C.c_int32 * 4), ("aw_cre", C.c_float), ("num_cp_max", C.c_int32), ("num_cp_min", C.c_int32),
                ("ratio_cp", C.c_float), ("thres_spd_cp", C.c_float), ("kernel", C.c_void_p * 3), ("kdim", (C.c_int32 * 2) * 3),
                ("seed", C.c_int64)]


_lib.mimc3_get_offset_image_multi.argtypes = [C.POINTER(_vp), C.c_int32, _f64p, C.c_int32, C.POINTER(CpParams), _i32p,
                                              np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.POINTER(C.c_int32), _i32p, _f32p]
_lib.mimc3_get_offset_image.argtypes = [_vp, _f64p, C.c_int32, C.POINTER(CpParams), _i32p,
                                        np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.POINTER(C.c_int32), _i32p, _f32p]


class VmapParams(C.Structure):
    """mimc3_vmap_params: the reference's `param_mimc2` + kernels (defaults = MIMC_main.c:134-194)."""
    _fields_ = [("vec_ocw", C.c_int32 * 4), ("aw_cre", C.c_float), ("aw_sf", C.c_float), ("radius_neighbor_dpf1", C.c_float),
                ("radius_neighbor_ps", C.c_float), ("num_cp_max", C.c_int32), ("num_cp_min", C.c_int32), ("ratio_cp", C.c_float),
                ("thres_spd_cp", C.c_float), ("kernel", C.c_void_p * 3), ("kdim", (C.c_int32 * 2) * 3), ("cp_seed", C.c_int64),
                ("qm_max_sweeps", C.c_int32)]


class VmapResult(C.Structure):
    _fields_ = [("dimx", C.c_int32), ("dimy", C.c_int32), ("mpp", C.c_float), ("spacing_grid", C.c_float),
                ("meter_per_spacing", C.c_float), ("cp_status", C.c_int32), ("offset_cp", C.c_int32 * 2), ("cp_subint", C.c_float * 2)]


CLI_KERNELS = (np.array([[-1, 0, 1]], np.float32), np.array([[-1], [0], [1]], np.float32),
               np.array([[-1 / 8] * 3, [-1 / 8, 1, -1 / 8], [-1 / 8] * 3], np.float32))     # MIMC_main.c:176-194

_lib.mimc3_postprocess.argtypes = [_vp, _f32p, C.c_int32, _f64p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int32, _f32p]
_lib.mimc3_vmap.argtypes = [_vp, _f64p, C.c_int32, C.c_float, C.POINTER(VmapParams), _f32p, _f32p, _f32p, _f32p, _f32p,
                            np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.POINTER(VmapResult)]
_lib.mimc3_vmap_passes.argtypes = [_vp, _f64p, C.c_int32, C.c_float, C.POINTER(VmapParams), C.c_int32, C.c_int32, _vp,
                                   np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.POINTER(VmapResult)]
_lib.mimc3_vmap_finish.argtypes = [_vp, _f64p, C.c_int32, C.c_float, C.POINTER(VmapParams), _vp, _f32p, _f32p, _f32p, _f32p, _f32p,
                                   C.POINTER(VmapResult)]
_lib.mimc3_vmap_geometry.argtypes = [_f64p, C.c_int32, C.POINTER(VmapResult)]
_lib.mimc3_vmap_cp.argtypes = [_vp, _f64p, C.c_int32, C.c_float, C.POINTER(VmapParams), np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"),
                               C.POINTER(VmapResult)]
_lib.mimc3_vmap_passes_points.argtypes = [_vp, _f64p, C.c_int32, C.c_float, C.POINTER(VmapParams), C.POINTER(VmapResult), _vp, C.c_int64]
_lib.mimc3_ctx_set_path.argtypes = [_vp, C.c_int32]
_lib.mimc3_ctx_last_path.argtypes = [_vp]
_lib.mimc3_ctx_enable_timing.argtypes = [_vp, C.c_int32]
_lib.mimc3_ctx_last_kernel_ms.argtypes = [_vp, C.POINTER(C.c_float)]


_lib.mimc3_point_cost.argtypes = [_i64p, C.c_int32, C.c_int32, _f64p]
_lib.mimc3_partition_points.argtypes = [_f64p, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, C.POINTER(C.c_double)]
_lib.mimc3_mgpu_create.argtypes = [_i32p, C.c_int32, C.POINTER(_vp)]
_lib.mimc3_mgpu_create_ex.argtypes = [_i32p, C.c_int32, C.c_char_p, C.c_uint32, C.POINTER(_vp)]
_lib.mimc3_mgpu_destroy.argtypes = [_vp]
_lib.mimc3_mgpu_destroy.restype = None
_lib.mimc3_mgpu_ndev.argtypes = [_vp]
_lib.mimc3_mgpu_ctx.argtypes = [_vp, C.c_int32]
_lib.mimc3_mgpu_ctx.restype = _vp
_lib.mimc3_mgpu_last_imbalance.argtypes = [_vp]
_lib.mimc3_mgpu_last_imbalance.restype = C.c_double
_lib.mimc3_mgpu_set_images.argtypes = [_vp, _f32p, _f32p, C.c_int32, C.c_int32]
_lib.mimc3_mgpu_set_images_u8.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32]
_lib.mimc3_mgpu_set_images_u16.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int32]
_lib.mimc3_mgpu_match_ncc_dlc.argtypes = [_vp, _f64p, C.c_int32, _i32p, _i32p, _i64p, C.c_int32, C.c_int32, _f32p]
_lib.mimc3_mgpu_vmap.argtypes = [_vp, _f64p, C.c_int32, C.c_float, C.POINTER(VmapParams), _f32p, _f32p, _f32p, _f32p, _f32p,
                                 np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.POINTER(VmapResult)]
_lib.mimc3_ctx_device.argtypes = [_vp]
_lib.mimc3_ctx_workspace.argtypes = [_vp, C.c_int32, C.c_size_t, C.POINTER(_vp)]


class Mimc3Error(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = _lib.mimc3_last_error().decode(errors="replace")
        super().__init__(f"{where}: rc={code}: {msg}")


def _check(rc, where):
    if rc != 0:
        raise Mimc3Error(rc, where)


def version():
    return _lib.mimc3_version().decode()


# ---------------------------------------------------------------------------------------------
# host-side geometry (no GPU needed)
# ---------------------------------------------------------------------------------------------
def get_uv_pivot(xyuvav, dt, mpp, ocw, H, W, aw_sf=1.8, aw_cre=10.0):
    """get_uv_pivot (MIMC_module.c:543-602) -> CSR (piv_off int64[N+1], piv_uv int32[P][2])."""
    xy = np.ascontiguousarray(xyuvav, np.float64)
    n = xy.shape[0]
    off = np.zeros(n + 1, np.int64)
    tot = C.c_int64(0)
    _check(_lib.mimc3_get_uv_pivot(xy, n, dt, mpp, aw_sf, aw_cre, ocw, H, W, off, None, 0, C.byref(tot)), "get_uv_pivot")
    uv = np.zeros((tot.value, 2), np.int32)
    _check(_lib.mimc3_get_uv_pivot(xy, n, dt, mpp, aw_sf, aw_cre, ocw, H, W, off, uv.ctypes.data_as(_vp), tot.value,
                                   C.byref(tot)), "get_uv_pivot")
    return off, uv


def get_uv_pivot_counts(xyuvav, dt, mpp, ocw, H, W, aw_sf=1.8, aw_cre=10.0):
    """The counting half of get_uv_pivot (MIMC_module.c:576-585): CSR offsets only (piv_off int64[N+1])."""
    xy = np.ascontiguousarray(xyuvav, np.float64)
    off = np.zeros(xy.shape[0] + 1, np.int64)
    tot = C.c_int64(0)
    _check(_lib.mimc3_get_uv_pivot(xy, xy.shape[0], dt, mpp, aw_sf, aw_cre, ocw, H, W, off, None, 0, C.byref(tot)), "get_uv_pivot")
    return off


CORRIDOR_BYTES = 24      # MIMC3_CORRIDOR_BYTES


def pivot_corridors(xyuvav, dt, mpp, aw_sf=1.8, aw_cre=10.0):
    """the host half of get_uv_pivot (MIMC_module.c:559-573): [N] opaque 24-byte corridor records (uint8 [N][24])"""
    xy = np.ascontiguousarray(xyuvav, np.float64)
    cor = np.zeros((xy.shape[0], CORRIDOR_BYTES), np.uint8)
    _check(_lib.mimc3_pivot_corridors(xy, xy.shape[0], dt, mpp, aw_sf, aw_cre, cor.ctypes.data), "pivot_corridors")
    return cor


def qm_launches_per_sweep():
    return int(_lib.mimc3_qm_launches_per_sweep())


def pivot_extent(piv_off, piv_uv):
    mn, mu, mv = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    off = np.ascontiguousarray(piv_off, np.int64)
    _check(_lib.mimc3_pivot_extent(np.ascontiguousarray(piv_uv, np.int32), off, off.shape[0] - 1, C.byref(mn), C.byref(mu),
                                   C.byref(mv)), "pivot_extent")
    return mn.value, mu.value, mv.value


def get_ruv_neighbor(xyuvav, dimx, dimy, meter_per_spacing, radius, cap=4096):
    """get_ruv_neighbor (MIMC_module.c:1266-1327) -> int32[nn][2]."""
    xy = np.ascontiguousarray(xyuvav, np.float64)
    ruv = np.zeros((cap, 2), np.int32)
    nn = C.c_int32(0)
    _check(_lib.mimc3_get_ruv_neighbor(xy, xy.shape[0], dimx, dimy, meter_per_spacing, radius, ruv, cap, C.byref(nn)),
           "get_ruv_neighbor")
    return np.ascontiguousarray(ruv[:nn.value])


def pinned_empty(shape, dtype):
    """numpy array in pinned host memory (mimc3_host_alloc): host<->device copies of it need no staging.
    The block lives as long as ANY view of it: numpy keeps the ctypes buffer object as the base of the array and of every
    slice taken from it, and the block is freed by a finalizer on that buffer object (not on the first array)."""
    import weakref
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dt.itemsize, 1)
    p = _lib.mimc3_host_alloc(nbytes)
    if not p:
        raise MemoryError("mimc3_host_alloc failed")
    buf = (C.c_char * nbytes).from_address(p)
    weakref.finalize(buf, _lib.mimc3_host_free, p)
    a = np.frombuffer(buf, dtype=dt, count=count).reshape(shape)
    a.flags.writeable = True
    return a


def point_cost(piv_off, ocw, cost=None):
    """mimc3_point_cost: adds (4 + 6 npiv)(2 ocw + 1)^2 per point to `cost` (float64[N], created when None)."""
    off = np.ascontiguousarray(piv_off, np.int64)
    n = off.shape[0] - 1
    if cost is None:
        cost = np.zeros(n, np.float64)
    _check(_lib.mimc3_point_cost(off, n, ocw, cost), "point_cost")
    return cost


def partition_points(cost, world, block=1024):
    """mimc3_partition_points: cost-balanced block-cyclic shares.  Returns (order int32[N], start int32[world+1],
    imbalance = max load / mean load - 1); rank r owns grid points order[start[r]:start[r+1]]."""
    cost = np.ascontiguousarray(cost, np.float64)
    n = cost.shape[0]
    order = np.empty(n, np.int32)
    start = np.empty(world + 1, np.int32)
    imb = C.c_double(0)
    _check(_lib.mimc3_partition_points(cost, n, world, block, order, start, C.byref(imb)), "partition_points")
    return order, start, imb.value


class MultiGpu:
    """mimc3_mgpu: ONE process driving several GPUs (a host thread per device, RCCL communicator over them)."""

    def __init__(self, devices, comm_lib=None, repeat_devices=False):
        """comm_lib / repeat_devices: the test form (mimc3_mgpu_create_ex) -- a stand-in communicator library, N ranks on one device."""
        self._h = _vp()
        dv = np.ascontiguousarray(devices, np.int32)
        if comm_lib or repeat_devices:
            _check(_lib.mimc3_mgpu_create_ex(dv, dv.shape[0], comm_lib.encode() if comm_lib else None, 1 if repeat_devices else 0, C.byref(self._h)), "mgpu_create_ex")
        else:
            _check(_lib.mimc3_mgpu_create(dv, dv.shape[0], C.byref(self._h)), "mgpu_create")
        self.ndev = dv.shape[0]

    def close(self):
        if self._h:
            _lib.mimc3_mgpu_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_images(self, i0, i1):
        if i0.dtype in (np.uint8, np.uint16):
            i0 = np.ascontiguousarray(i0); i1 = np.ascontiguousarray(i1)
            fn = _lib.mimc3_mgpu_set_images_u8 if i0.dtype == np.uint8 else _lib.mimc3_mgpu_set_images_u16
            _check(fn(self._h, i0.ctypes.data, i1.ctypes.data, i0.shape[0], i0.shape[1]), "mgpu_set_images_raw")
        else:
            i0 = np.ascontiguousarray(i0, np.float32); i1 = np.ascontiguousarray(i1, np.float32)
            _check(_lib.mimc3_mgpu_set_images(self._h, i0, i1, i0.shape[0], i0.shape[1]), "mgpu_set_images")

    def matching_ncc_dlc_2(self, xyuvav, offset, piv_off, piv_uv, ocw, swap=False):
        xy = np.ascontiguousarray(xyuvav, np.float64)
        out = np.empty((xy.shape[0], 3), np.float32)
        _check(_lib.mimc3_mgpu_match_ncc_dlc(self._h, xy, xy.shape[0], np.ascontiguousarray(offset, np.int32),
                                             np.ascontiguousarray(piv_uv, np.int32), np.ascontiguousarray(piv_off, np.int64), ocw,
                                             1 if swap else 0, out), "mgpu_matching_ncc_dlc_2")
        return out

    def vmap(self, xyuvav, dt, **kw):
        xy = np.ascontiguousarray(xyuvav, np.float64)
        n = xy.shape[0]
        p, _keep = Context._vmap_params(**kw)
        planes = [np.empty(n, np.float32) for _ in range(5)]
        flag = np.zeros(n, np.uint8)
        r = VmapResult()
        _check(_lib.mimc3_mgpu_vmap(self._h, xy, n, dt, C.byref(p), *planes, flag, C.byref(r)), "mgpu_vmap")
        return Context._vmap_out(r, flag, planes)

    def last_imbalance(self):
        return float(_lib.mimc3_mgpu_last_imbalance(self._h))


# ---------------------------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------------------------
class Context:
    """Owns a HIP stream and the resident image pair (mimc3_ctx)."""

    def __init__(self, device=0):
        self._h = _vp()
        _check(_lib.mimc3_ctx_create(device, C.byref(self._h)), "ctx_create")
        self.device = device
        self.H = self.W = 0
        self._keep = None

    def close(self):
        if self._h:
            _lib.mimc3_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- images -------------------------------------------------------------------------------
    def set_images(self, i0, i1):
        i0 = np.ascontiguousarray(i0, np.float32)
        i1 = np.ascontiguousarray(i1, np.float32)
        assert i0.shape == i1.shape and i0.ndim == 2
        self.H, self.W = i0.shape
        _check(_lib.mimc3_ctx_set_images(self._h, i0, i1, self.H, self.W), "ctx_set_images")

    def set_images_raw(self, i0, i1):
        """The pair as the TIFF holds it (uint8 or uint16 arrays): raw DN crosses PCIe, the widening to float32 of
        GMA_float_load_tiff (GMA.c:288-310) runs on the device.  Arrays from pinned_empty() are DMA'd without staging."""
        assert i0.shape == i1.shape and i0.ndim == 2 and i0.dtype == i1.dtype and i0.dtype in (np.uint8, np.uint16)
        i0 = np.ascontiguousarray(i0); i1 = np.ascontiguousarray(i1)
        self.H, self.W = i0.shape
        fn = _lib.mimc3_ctx_set_images_u8 if i0.dtype == np.uint8 else _lib.mimc3_ctx_set_images_u16
        _check(fn(self._h, i0.ctypes.data, i1.ctypes.data, self.H, self.W), "ctx_set_images_raw")

    def set_images_dev(self, d_i0, d_i1, H, W, keep=None):
        """d_i0/d_i1: integer device addresses (e.g. torch tensor.data_ptr()); keep = objects to hold."""
        self.H, self.W = H, W
        self._keep = keep
        _check(_lib.mimc3_ctx_set_images_dev(self._h, d_i0, d_i1, H, W), "ctx_set_images_dev")

    # -- matcher ------------------------------------------------------------------------------
    def matching_ncc_dlc_2(self, xyuvav, offset, piv_off, piv_uv, ocw, swap=False):
        """matching_ncc_dlc_2 (MIMC_module.c:805-842) on the resident pair -> float32[N][3].
        swap=True = the CLI's "swapped forward" pass (chip from i1, window from i0)."""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        n = xy.shape[0]
        out = np.empty((n, 3), np.float32)
        _check(_lib.mimc3_match_ncc_dlc(self._h, xy, n, np.ascontiguousarray(offset, np.int32),
                                        np.ascontiguousarray(piv_uv, np.int32), np.ascontiguousarray(piv_off, np.int64),
                                        ocw, 1 if swap else 0, out), "matching_ncc_dlc_2")
        return out

    def matching_ncc_dlc_geo(self, xyuvav, offset, dt, mpp, ocw, swap=False, aw_sf=1.8, aw_cre=10.0, out=None):
        """get_uv_pivot + matching_ncc_dlc_2 in one call: corridors from the host (made chunk by chunk under the device's work), pivot
        lists made on the device.  swap=True = the swapped pass (images exchanged, pivots negated; the caller negates offset and
        (du, dv)).  `out` may be a pinned array (as for matching_ncc_dlc_cor)."""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        if out is None:
            out = np.empty((xy.shape[0], 3), np.float32)
        _check(_lib.mimc3_match_ncc_dlc_geo(self._h, xy, xy.shape[0], np.ascontiguousarray(offset, np.int32), dt, mpp, aw_sf, aw_cre, ocw,
                                            1 if swap else 0, out), "matching_ncc_dlc_geo")
        return out

    def matching_ncc_dlc_cor(self, xyuvav, cor, offset, ocw, swap=False, out=None):
        """the same with the corridors given (pivot_corridors(): once per grid, whatever the chip size); `out` may be a pinned array"""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        if out is None:
            out = np.empty((xy.shape[0], 3), np.float32)
        cor = np.ascontiguousarray(cor, np.uint8)
        _check(_lib.mimc3_match_ncc_dlc_cor(self._h, xy, cor.ctypes.data, xy.shape[0], np.ascontiguousarray(offset, np.int32), ocw, 1 if swap else 0, out),
               "matching_ncc_dlc_cor")
        return out

    def get_uv_pivot_dev(self, d_xyuvav, d_cor, n, ocw, d_piv_off, d_piv_uv=None, d_piv_uv_neg=None, cap=0, stream=0):
        """device half of get_uv_pivot on device buffers; returns (total, (max_npiv, max_abs_u, max_abs_v))"""
        total = C.c_int64(0)
        ext = np.zeros(3, np.int32)
        _check(_lib.mimc3_get_uv_pivot_dev(self._h, d_xyuvav, d_cor, n, ocw, d_piv_off, d_piv_uv, d_piv_uv_neg, cap, C.byref(total), ext, stream),
               "get_uv_pivot_dev")
        return total.value, (int(ext[0]), int(ext[1]), int(ext[2]))

    def matching_ncc_dlc_2_dev(self, d_xyuvav, n, offset, d_piv_uv, d_piv_off, extent, ocw, d_out, stream=0, swap=False):
        """Device-pointer variant (enqueue only). extent = pivot_extent(...) = (max_npiv, max|u|, max|v|)."""
        mn, mu, mv = extent
        _check(_lib.mimc3_match_ncc_dlc_dev(self._h, d_xyuvav, n, int(offset[0]), int(offset[1]), d_piv_uv, d_piv_off,
                                            mn, mu, mv, ocw, 1 if swap else 0, d_out, stream), "matching_ncc_dlc_2_dev")

    # -- QM -----------------------------------------------------------------------------------
    def get_dpf_pseudosmoothing(self, dpf, dpf_dx, dpf_dy, ruv, mvn, nclus, xyuvav, max_sweeps=101):
        """get_dpf_pseudosmoothing (MIMC_module.c:1986-2312). Returns (dpf, dx, dy, sweeps); inputs untouched."""
        dimy, dimx = dpf.shape
        d = np.array(dpf, np.int32, order="C")
        x = np.array(dpf_dx, np.float32, order="C")
        y = np.array(dpf_dy, np.float32, order="C")
        ruv = np.ascontiguousarray(ruv, np.int32)
        mvn = np.ascontiguousarray(mvn, np.float32)
        sw = C.c_int32(0)
        _check(_lib.mimc3_qm_pseudosmooth(self._h, dimy, dimx, d.reshape(-1), x.reshape(-1), y.reshape(-1), ruv,
                                          ruv.shape[0], mvn, mvn.shape[1], np.ascontiguousarray(nclus, np.int32),
                                          np.ascontiguousarray(xyuvav, np.float64), max_sweeps, C.byref(sw)),
               "get_dpf_pseudosmoothing")
        return d, x, y, sw.value

    def qm_workspace_bytes(self, ngrid, max_sweeps):
        return int(_lib.mimc3_qm_workspace_bytes(ngrid, max_sweeps))

    def get_dpf_pseudosmoothing_dev(self, dimy, dimx, d_dpf, d_dx, d_dy, d_ruv, nn, d_mvn, kmax, d_nclus, d_xyuvav,
                                    max_sweeps, d_work, d_sweeps=None, stream=0):
        _check(_lib.mimc3_qm_pseudosmooth_dev(self._h, dimy, dimx, d_dpf, d_dx, d_dy, d_ruv, nn, d_mvn, kmax, d_nclus,
                                              d_xyuvav, max_sweeps, d_work, d_sweeps, stream), "get_dpf_pseudosmoothing_dev")

    # -- N3: the program's data path on arrays -----------------------------------------------------
    def mimc2_postprocess(self, dp, xyuvav, dimx, dimy, dt, mpp, meter_per_spacing, radius_dpf1=3.0, radius_ps=5.0,
                          qm_max_sweeps=101):
        """mimc2_postprocess (MIMC_module.c:892-990): dp [ndp][N][3] -> vxyexyqual [5][dimy][dimx] (px units)."""
        dp = np.ascontiguousarray(dp, np.float32)
        out = np.empty((5, dimy, dimx), np.float32)
        _check(_lib.mimc3_postprocess(self._h, dp, dp.shape[0], np.ascontiguousarray(xyuvav, np.float64), dimx, dimy, dt, mpp,
                                      meter_per_spacing, radius_dpf1, radius_ps, qm_max_sweeps, out.reshape(-1)), "mimc2_postprocess")
        return out

    @staticmethod
    def _vmap_params(kernels=CLI_KERNELS, cp_seed=-1, vec_ocw=(7, 15, 30, 40), aw_cre=10.0, aw_sf=1.8, radius_neighbor_dpf1=3.0,
                     radius_neighbor_ps=5.0, num_cp_max=500, num_cp_min=50, ratio_cp=0.03, thres_spd_cp=10.0, qm_max_sweeps=101):
        ks = [np.ascontiguousarray(k, np.float32) for k in kernels]
        p = VmapParams()
        p.vec_ocw[:] = list(vec_ocw)
        p.aw_cre = aw_cre; p.aw_sf = aw_sf; p.radius_neighbor_dpf1 = radius_neighbor_dpf1; p.radius_neighbor_ps = radius_neighbor_ps
        p.num_cp_max = num_cp_max; p.num_cp_min = num_cp_min; p.ratio_cp = ratio_cp; p.thres_spd_cp = thres_spd_cp
        p.cp_seed = cp_seed; p.qm_max_sweeps = qm_max_sweeps
        for i, k in enumerate(ks):
            p.kernel[i] = k.ctypes.data
            p.kdim[i][0], p.kdim[i][1] = k.shape
        return p, ks                      # ks keeps the kernel arrays alive

    @staticmethod
    def _vmap_out(r, flag, planes):
        out = dict(dimx=r.dimx, dimy=r.dimy, mpp=r.mpp, spacing_grid=r.spacing_grid, meter_per_spacing=r.meter_per_spacing,
                   cp_status=r.cp_status, offset_cp=(r.offset_cp[0], r.offset_cp[1]), cp_subint=(r.cp_subint[0], r.cp_subint[1]),
                   flag_cp=flag)
        for name, a in zip(("vx", "vy", "ex", "ey", "qual"), planes):
            out[name] = a.reshape(r.dimy, r.dimx) if r.cp_status > 0 else None
        return out

    def vmap(self, xyuvav, dt, **kw):
        """MIMC_main.c:203-402 on the resident pair (keywords: see _vmap_params; defaults = MIMC_main.c:134-194).
        Returns dict(vx, vy, ex, ey, qual [dimy][dimx], flag_cp, + the scalar fields of mimc3_vmap_result); the planes
        are None when cp_status == -1."""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        n = xy.shape[0]
        p, _keep = self._vmap_params(**kw)
        planes = [np.empty(n, np.float32) for _ in range(5)]
        flag = np.zeros(n, np.uint8)
        r = VmapResult()
        _check(_lib.mimc3_vmap(self._h, xy, n, dt, C.byref(p), *planes, flag, C.byref(r)), "vmap")
        return self._vmap_out(r, flag, planes)

    def vmap_passes(self, xyuvav, dt, lo, hi, d_dp, **kw):
        """mimc3_vmap_passes: CP offset on the whole grid + the 32 passes for grid points [lo, hi) into the device tensor
        d_dp [32][hi-lo][3] (pass its data_ptr()).  Returns (VmapResult, flag_cp)."""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        p, _keep = self._vmap_params(**kw)
        flag = np.zeros(xy.shape[0], np.uint8)
        r = VmapResult()
        _check(_lib.mimc3_vmap_passes(self._h, xy, xy.shape[0], dt, C.byref(p), lo, hi, d_dp, flag, C.byref(r)), "vmap_passes")
        return r, flag

    def vmap_geometry(self, xyuvav):
        xy = np.ascontiguousarray(xyuvav, np.float64)
        r = VmapResult()
        _check(_lib.mimc3_vmap_geometry(xy, xy.shape[0], C.byref(r)), "vmap_geometry")
        return r

    def vmap_cp(self, xyuvav, dt, **kw):
        """geometry + CP offset on the whole grid -> (VmapResult, flag_cp)"""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        p, _keep = self._vmap_params(**kw)
        flag = np.zeros(xy.shape[0], np.uint8)
        r = VmapResult()
        _check(_lib.mimc3_vmap_cp(self._h, xy, xy.shape[0], dt, C.byref(p), flag, C.byref(r)), "vmap_cp")
        return r, flag

    def vmap_passes_points(self, xs, dt, r, d_dp, pass_stride=0, **kw):
        """the 32 passes for the grid points xs [n][6] (any subset) into the device tensor d_dp [32][pass_stride][3]"""
        xs = np.ascontiguousarray(xs, np.float64)
        p, _keep = self._vmap_params(**kw)
        _check(_lib.mimc3_vmap_passes_points(self._h, xs, xs.shape[0], dt, C.byref(p), C.byref(r), d_dp, pass_stride), "vmap_passes_points")

    def vmap_finish(self, xyuvav, dt, d_dp_full, r, flag, **kw):
        """mimc3_vmap_finish on the complete candidate tensor [32][N][3] (device pointer); same dict as vmap()."""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        n = xy.shape[0]
        p, _keep = self._vmap_params(**kw)
        planes = [np.empty(n, np.float32) for _ in range(5)]
        if r.cp_status > 0:
            _check(_lib.mimc3_vmap_finish(self._h, xy, n, dt, C.byref(p), d_dp_full, *planes, C.byref(r)), "vmap_finish")
        return self._vmap_out(r, flag, planes)

    # -- N4: control-point offset -----------------------------------------------------------------
    def get_offset_image(self, xyuvav, kernels, seed=-1, vec_ocw=(7, 15, 30, 40), aw_cre=10.0, num_cp_max=500, num_cp_min=50,
                         ratio_cp=0.03, thres_spd_cp=10.0, peers=()):
        """get_offset_image (MIMC_module.c:33-492) on the resident pair. Defaults = MIMC_main.c:134-170.
        Returns (status, offset[2], flag_cp[N], info[4], sduv[2]); status 1 = ok, -1 = not enough control points.
        peers: further Contexts holding the SAME pair (other GPUs) that share the device work (mimc3_get_offset_image_multi);
        the result does not depend on them."""
        xy = np.ascontiguousarray(xyuvav, np.float64)
        ks = [np.ascontiguousarray(k, np.float32) for k in kernels]
        p = CpParams()
        p.vec_ocw[:] = list(vec_ocw)
        p.aw_cre = aw_cre; p.num_cp_max = num_cp_max; p.num_cp_min = num_cp_min
        p.ratio_cp = ratio_cp; p.thres_spd_cp = thres_spd_cp; p.seed = seed
        for i, k in enumerate(ks):
            p.kernel[i] = k.ctypes.data
            p.kdim[i][0], p.kdim[i][1] = k.shape
        off = np.zeros(2, np.int32)
        flag = np.zeros(xy.shape[0], np.uint8)
        info = np.zeros(4, np.int32)
        sduv = np.zeros(2, np.float32)
        st = C.c_int32(0)
        if peers:
            hs = (_vp * (1 + len(peers)))(self._h, *[q._h for q in peers])
            _check(_lib.mimc3_get_offset_image_multi(hs, len(hs), xy, xy.shape[0], C.byref(p), off, flag, C.byref(st), info, sduv),
                   "get_offset_image")
        else:
            _check(_lib.mimc3_get_offset_image(self._h, xy, xy.shape[0], C.byref(p), off, flag, C.byref(st), info, sduv),
                   "get_offset_image")
        return st.value, off, flag, info, sduv

    # -- N2: image pre-filter ---------------------------------------------------------------------
    def GMA_float_conv2(self, img, kernel, out=None):
        """GMA_float_conv2 (MIMC_module.c:2517-2585). `out` is in/out as in the reference (its border takes part in
        the minimum); default = a zero plane, which is what a fresh GMA_float_create gives the CLI."""
        img = np.ascontiguousarray(img, np.float32)
        kernel = np.ascontiguousarray(kernel, np.float32)
        o = np.zeros_like(img) if out is None else np.array(out, np.float32, order="C")
        _check(_lib.mimc3_float_conv2(self._h, img, img.shape[0], img.shape[1], kernel, kernel.shape[0], kernel.shape[1], o),
               "GMA_float_conv2")
        return o

    def GMA_float_conv2_dev(self, d_in, H, W, kernel, d_out, d_scratch, stream=0):
        kernel = np.ascontiguousarray(kernel, np.float32)
        _check(_lib.mimc3_float_conv2_dev(self._h, d_in, H, W, kernel, kernel.shape[0], kernel.shape[1], d_out, d_scratch, stream),
               "GMA_float_conv2_dev")

    def filter_images(self, kernel):
        """Filter the resident pair on the device and match on the filtered pair from now on; None = back to raw."""
        if kernel is None:
            _check(_lib.mimc3_ctx_filter_images(self._h, None, 0, 0), "filter_images")
            return
        k = np.ascontiguousarray(kernel, np.float32)
        _check(_lib.mimc3_ctx_filter_images(self._h, k.ctypes.data_as(_vp), k.shape[0], k.shape[1]), "filter_images")

    def get_images(self, H, W):
        i0 = np.empty((H, W), np.float32); i1 = np.empty((H, W), np.float32)
        _check(_lib.mimc3_ctx_get_images(self._h, i0.ctypes.data_as(_vp), i1.ctypes.data_as(_vp)), "get_images")
        return i0, i1

    # -- N1: clustering, dpf0, dpf1 -------------------------------------------------------------
    def calc_mean_var_num_dp_cluster(self, dp, kmax=None):
        """calc_mean_var_num_dp_cluster (MIMC_module.c:994-1130). dp [ndp][N][3] -> (mvn [N][kmax][5], nclus [N]).
        kmax defaults to ndp (always enough); raises Mimc3Error(ECAP) if a point has more clusters than kmax."""
        dp = np.ascontiguousarray(dp, np.float32)
        ndp, n, _ = dp.shape
        kmax = int(kmax or ndp)
        mvn = np.empty((n, kmax, 5), np.float32)
        nclus = np.empty(n, np.int32)
        seen = C.c_int32(0)
        _check(_lib.mimc3_cluster_candidates(self._h, dp, ndp, n, kmax, mvn, nclus, C.byref(seen)),
               "calc_mean_var_num_dp_cluster")
        return mvn, nclus

    def calc_mean_var_num_dp_cluster_dev(self, d_dp, ndp, n, kmax, d_mvn, d_nclus, d_kmax_seen, stream=0):
        _check(_lib.mimc3_cluster_candidates_dev(self._h, d_dp, ndp, n, kmax, d_mvn, d_nclus, d_kmax_seen, stream),
               "calc_mean_var_num_dp_cluster_dev")

    def get_dpf0(self, mvn, nclus, dimx, dimy, min_ratio=0.6):
        """get_dpf0 (MIMC_module.c:1224-1263) -> dpf0 [dimy][dimx]."""
        mvn = np.ascontiguousarray(mvn, np.float32)
        dpf = np.empty(dimx * dimy, np.int32)
        _check(_lib.mimc3_get_dpf0(self._h, mvn, np.ascontiguousarray(nclus, np.int32), dimx * dimy, mvn.shape[1],
                                   min_ratio, dpf), "get_dpf0")
        return dpf.reshape(dimy, dimx)

    def get_dpf0_dev(self, d_mvn, d_nclus, n, kmax, min_ratio, d_dpf, stream=0):
        _check(_lib.mimc3_get_dpf0_dev(self._h, d_mvn, d_nclus, n, kmax, min_ratio, d_dpf, stream), "get_dpf0_dev")

    def get_dpf1(self, dpf0, ruv, mvn, nclus, xyuvav, dt, mpp):
        """get_dpf1 (MIMC_module.c:1330-1718). Returns (dpf1, dx, dy, sweeps); inputs untouched."""
        dimy, dimx = dpf0.shape
        d = np.array(dpf0, np.int32, order="C")
        x = np.empty((dimy, dimx), np.float32)
        y = np.empty((dimy, dimx), np.float32)
        ruv = np.ascontiguousarray(ruv, np.int32)
        mvn = np.ascontiguousarray(mvn, np.float32)
        sw = C.c_int32(0)
        _check(_lib.mimc3_get_dpf1(self._h, dimy, dimx, d.reshape(-1), x.reshape(-1), y.reshape(-1), ruv, ruv.shape[0],
                                   mvn, mvn.shape[1], np.ascontiguousarray(nclus, np.int32),
                                   np.ascontiguousarray(xyuvav, np.float64), dt, mpp, C.byref(sw)), "get_dpf1")
        return d, x, y, sw.value

    def dpf1_workspace_bytes(self, ngrid):
        return int(_lib.mimc3_dpf1_workspace_bytes(ngrid))

    def get_dpf1_dev(self, dimy, dimx, d_dpf, d_dx, d_dy, d_ruv, nn, d_mvn, kmax, d_nclus, d_xyuvav, dt, mpp, d_work,
                     stream=0):
        """Device-resident get_dpf1; synchronises `stream` once per 32 sweeps. Returns the sweep count."""
        sw = C.c_int32(0)
        _check(_lib.mimc3_get_dpf1_dev(self._h, dimy, dimx, d_dpf, d_dx, d_dy, d_ruv, nn, d_mvn, kmax, d_nclus, d_xyuvav,
                                       dt, mpp, d_work, C.byref(sw), stream), "get_dpf1_dev")
        return sw.value

    # -- kernel selection ---------------------------------------------------------------------
    def set_path(self, mode):
        """"auto" (0): u8 kernel when the pair is 8-bit integral, else the tiled f32 kernel, else the general
        one; "general" (1): force the general f32 kernel; "f32" (2): like auto but never the u8 kernel."""
        _check(_lib.mimc3_ctx_set_path(self._h, {"auto": 0, "general": 1, "f32": 2, "u16": 3, "u8px": 4}.get(mode, mode)), "set_path")

    def last_path(self):
        return {0: "general_f32", 1: "u8_exact", 2: "f32_tiled", 3: "u16_scaled", 4: "u8_offset", 5: "u8_mfma"}.get(int(_lib.mimc3_ctx_last_path(self._h)), "none")

    # -- timing -------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        _check(_lib.mimc3_ctx_enable_timing(self._h, 1 if on else 0), "enable_timing")

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(_lib.mimc3_ctx_last_kernel_ms(self._h, C.byref(ms)), "last_kernel_ms")
        return float(ms.value)
