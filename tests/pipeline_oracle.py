"""Test infrastructure: the reference program's data path (MIMC_main.c:203-402) chained from the ORACLE's functions,
for checking mimc3_vmap / the CLI where the compiled reference program is not available (the GPU box)."""
import numpy as np

CLI_KERNELS = (np.array([[-1, 0, 1]], np.float32), np.array([[-1], [0], [1]], np.float32),
               np.array([[-1 / 8] * 3, [-1 / 8, 1, -1 / 8], [-1 / 8] * 3], np.float32))


def oracle_vmap(orc, i0, i1, xyuvav, dt, cp_seed, vec_ocw=(7, 15, 30, 40), aw_cre=10.0, aw_sf=1.8, radius_dpf1=3.0,
                radius_ps=5.0, num_cp_max=500, num_cp_min=50, ratio_cp=0.03, thres_spd_cp=10.0, kernels=CLI_KERNELS):
    xy = np.ascontiguousarray(xyuvav, np.float64)
    n = xy.shape[0]
    H, W = i0.shape
    g = 1
    while g < n and int(xy[g, 2]) != int(xy[0, 2]):
        g += 1
    dimx, dimy = g, n // g
    mpp = np.float32((xy[1, 0] - xy[0, 0]) / (xy[1, 2] - xy[0, 2]))
    mps = np.float32(xy[1, 0] - xy[0, 0])
    rc, off, flag, _, _ = orc.get_offset_image(i0, i1, xy, kernels, cp_seed, vec_ocw=vec_ocw, aw_cre=aw_cre, num_cp_max=num_cp_max,
                                               num_cp_min=num_cp_min, ratio_cp=ratio_cp, thres_spd_cp=thres_spd_cp)
    out = dict(dimx=dimx, dimy=dimy, mpp=float(mpp), cp_status=rc, offset_cp=(int(off[0]), int(off[1])), flag_cp=flag)
    if rc < 0:
        return out
    dp = np.zeros((32, n, 3), np.float32)
    a0, a1 = i0, i1
    c0 = np.zeros_like(i0); c1 = np.zeros_like(i1)              # i0c, i1c: created once (MIMC_main.c:302-303)
    for kk in range(-1, 3):
        if kk >= 0:
            c0 = orc.float_conv2(i0, kernels[kk], c0); c1 = orc.float_conv2(i1, kernels[kk], c1)
            a0, a1 = c0, c1
        for c, ocw in enumerate(vec_ocw):
            po, pu = orc.get_uv_pivot(xy, dt, float(mpp), ocw, H, W, aw_sf, aw_cre)
            slot = (kk + 1) * 8 + 2 * c
            dp[slot] = orc.match(a0, a1, xy, off, po, pu, ocw)
            sw = orc.match(a1, a0, xy, -off, po, -pu, ocw)
            sw[:, :2] = -sw[:, :2]
            dp[slot + 1] = sw
    mvn, nclus = orc.cluster_candidates(dp, kmax=32)
    d0 = orc.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
    ruv1 = orc.get_ruv_neighbor(xy, dimx, dimy, float(mps), radius_dpf1)
    d1, x1, y1 = orc.get_dpf1(d0, ruv1, mvn, nclus, xy, dt, float(mpp))
    ruv2 = orc.get_ruv_neighbor(xy, dimx, dimy, float(mps), radius_ps)
    d2 = orc.qm(d1, x1, y1, ruv2, mvn, nclus, xy)[0].reshape(-1)
    planes = np.full((5, n), np.nan, np.float32)
    ok = d2 >= 0
    planes[:, ok] = mvn[np.arange(n)[ok], d2[ok], :].T
    vx, vy, ex, ey, qual = planes
    sdu = np.float32(0); sdv = np.float32(0); num = 0
    for a, b in zip(vx, vy):                                    # f32 running sums in grid order (:362-376)
        if not (np.isnan(a) or np.isnan(b)):
            sdu = np.float32(sdu + a); sdv = np.float32(sdv + b); num += 1
    du = np.float32(sdu / np.float32(num)); dv = np.float32(sdv / np.float32(num))
    factor = np.float32(np.float32(mpp / np.float32(dt)) * np.float32(365))
    vx = (vx - du) * factor
    vy = -(vy - dv) * factor
    with np.errstate(invalid="ignore"):
        ex = (np.sqrt(ex.astype(np.float64)) * np.float64(factor)).astype(np.float32)
        ey = (np.sqrt(ey.astype(np.float64)) * np.float64(factor)).astype(np.float32)
    out.update(vx=vx.reshape(dimy, dimx), vy=vy.reshape(dimy, dimx), ex=ex.reshape(dimy, dimx), ey=ey.reshape(dimy, dimx),
               qual=qual.reshape(dimy, dimx), cp_subint=(float(du), float(dv)), dp=dp)
    return out
