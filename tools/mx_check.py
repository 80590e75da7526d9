#!/usr/bin/env python3
"""Matrix-core matcher (match_mx_kernel.hip) against the register-tiled u8 kernel, bit for bit, on the GPU box:
    python3 tools/mx_check.py [C2] [small] [--ocw 16,7,...]
For each case: forward and swapped pass in both modes ("auto" = matrix-core kernel first, "u8px" = register-tiled only),
the number of differing output words, and the kernel times (HIP events through the context's timing hooks).
Tuning / bring-up infrastructure; the parity tests proper are under tests/."""
import json
import os
import sys

import numpy as np
import torch   # before mimc3_amd.api (one HIP runtime per process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimc3_amd import api, synth  # noqa: E402


def bits_diff(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    ai = np.where(na, 0, a).view(np.uint32); bi = np.where(nb, 0, b).view(np.uint32)
    bad = (ai != bi) | (na != nb)
    return int(bad.any(axis=1).sum()), np.argwhere(bad.any(axis=1))[:5, 0].tolist()


def run_case(name, c, ocws, reps=3):
    H, W = c.i0.shape
    for ocw in ocws:
        off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
        res = {}
        with api.Context(0) as ctx:
            ctx.set_images(c.i0, c.i1)
            ctx.enable_timing(True)
            for mode in ("u8px", "auto"):
                ctx.set_path(mode)
                ms = []
                for r in range(reps):
                    fw = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
                    ms.append(ctx.last_kernel_ms())
                sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
                res[mode] = (fw, sw, min(ms), ctx.last_path())
        nf, wf = bits_diff(res["auto"][0], res["u8px"][0])
        ns, ws = bits_diff(res["auto"][1], res["u8px"][1])
        print(json.dumps({"case": name, "ocw": ocw, "points": int(c.xyuvav.shape[0]), "paths": [res["u8px"][3], res["auto"][3]],
                          "diff_forward": nf, "diff_swapped": ns, "first_bad": wf + ws,
                          "ms_u8px": round(res["u8px"][2], 3), "ms_auto": round(res["auto"][2], 3),
                          "invalid": int(np.isnan(res["auto"][0][:, 0]).sum())}), flush=True)
        if nf and os.environ.get("MX_CHECK_VERBOSE"):
            for g in wf[:6]:
                u0, v0 = int(c.xyuvav[g, 2]), int(c.xyuvav[g, 3])
                chip = c.i0[v0 - ocw:v0 + ocw + 1, u0 - ocw:u0 + ocw + 1]
                k0, k1 = int(off[g]), int(off[g + 1])
                lu, lv = int(uv[k1 - 1, 0]), int(uv[k1 - 1, 1])
                dx2, dy2 = abs(lu) + ocw + 2, abs(lv) + ocw + 2
                wu, wv = u0 + int(c.offset[0]) - dx2, v0 + int(c.offset[1]) - dy2
                win = c.i1[max(wv, 0):wv + 2 * dy2, max(wu, 0):wu + 2 * dx2]
                print("   point", g, "npiv", k1 - k0, "chip nulls", int((chip == 0).sum()), "chip null rows", int((chip == 0).any(axis=1).sum()),
                      "window nulls", int((win == 0).sum()), "auto", res["auto"][0][g].tolist(), "u8px", res["u8px"][0][g].tolist(), flush=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ocws = [16]
    if "--ocw" in sys.argv:
        ocws = [int(x) for x in sys.argv[sys.argv.index("--ocw") + 1].split(",")]
        args = [a for a in args if "," not in a and not a.isdigit()]
    if not args:
        args = ["small", "C2"]
    for a in args:
        if a == "small":
            cases = [dict(seed=35, shift=(6, -6), angle_deg=45.0, speed=1806.0, h=260, w=260, noise_dn=2, null_frac=0.03),
                     dict(seed=41, shift=(2, 2), angle_deg=-40.0, speed=1500.0, h=240, w=250, offset=(-1, 2)),
                     dict(seed=42, shift=(-3, 1), angle_deg=170.0, speed=1000.0, h=230, w=260, noise_dn=3, null_frac=0.10),
                     dict(seed=43, shift=(0, 4), angle_deg=-88.0, speed=1900.0, h=300, w=240, null_frac=0.30),
                     dict(seed=44, shift=(4, -3), angle_deg=38.0, speed=2600.0, h=300, w=300, noise_dn=2, null_frac=0.05, offset=(3, -2))]
            for k in cases:
                for ocw in ocws:
                    kk = dict(k); kk["ocw"] = ocw
                    run_case("small_%d" % k["seed"], synth.make_small(**kk), [ocw])
        else:
            run_case(a, synth.make_case(a), ocws)


if __name__ == "__main__":
    main()
