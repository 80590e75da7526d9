#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: grid-points/s of the DLC/NCC matcher.

    python bench.py --gpus N --steps K --warmup W           (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one matcher pass (matching_ncc_dlc_2, MIMC_module.c:805-842) over one batch of synthetic grid points.
Workload at N=1 = BASELINE configs[1] (C2): 4096x4096 synthetic pair, 200,000 grid points (500x400), ocw 16 (33x33 chip),
~15 DLC pivots (65x65 window).  Inputs (both images, xyuvav, pivot CSR) are resident in HBM before the timed region:
`value` is the kernel-only rate; `value_incl_io` is SURVEY 8(d)(i)'s figure with the pivot/xyuvav upload and the result
download inside every step.

At N>1 (one rank per GPU, torch.distributed, backend nccl = RCCL) the headline is BASELINE configs[2] (C3): the SAME
200,000 points sharded over the ranks in cost-balanced blocks (strong scaling), every step = the rank's share + ONE
all-gather of the padded (du,dv,ncc) blocks + the re-ordering to grid order.  The weak-scaling figure (every rank matches
its own 200,000-point lattice) is reported next to it in `weak`.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and, at N=1, `cpu_baseline` (the compiled
reference OpenMP path when oracle/_ref travelled here, else the parity-verified C restatement), a parity figure against
it, and `f32_path` (the same workload forced onto the tiled f32 kernel, what 16-bit imagery gets).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)
# (f32_tiled: PxF32i is the tiled f32 kernel on integral pixels; PxF32 otherwise)
KERNEL_NAMES = {"u8_mfma": "match_ncc_dlc_mx", "u8_exact": "match_ncc_dlc_px<PxU8>", "f32_tiled": "match_ncc_dlc_px<PxF32i>",
                "u16_scaled": "match_ncc_dlc_px<PxU16>", "u8_offset": "match_ncc_dlc_px<PxU8o>", "general_f32": "match_ncc_dlc_f32"}
DTYPES = {"u8_mfma": "u8", "u8_exact": "u8", "u16_scaled": "u16", "u8_offset": "u8"}


def kernel_source_sha16():
    """identifies the matcher kernel source the PMC figures in profiles/traffic_latest.json were measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("match_mx_kernel.hip", "match_px_kernel.hip", "match_kernel.h", "sat_kernel.h"):
        with open(os.path.join(ROOT, "mimc3_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(piv_off, piv_uv, ocw):
    """B = 4(2ocw+1)^2 + 4*Dy2*Dx2 + 48 + 8*npiv + 12 per grid point, summed."""
    npiv = (piv_off[1:] - piv_off[:-1]).astype(np.int64)
    last = piv_uv[piv_off[1:] - 1].astype(np.int64)
    dx2 = np.abs(last[:, 0]) + ocw + 2
    dy2 = np.abs(last[:, 1]) + ocw + 2
    chip = 4 * (2 * ocw + 1) ** 2
    return int((chip + 4 * (2 * dx2 + 1) * (2 * dy2 + 1) + 48 + 8 * npiv + 12).sum())


class Leg:
    """one resident matcher workload on this rank: device copies of xyuvav + pivot CSR, an output block of `per` rows"""

    def __init__(self, torch, api, dev, xy, piv_off, piv_uv, per):
        self.n = xy.shape[0]
        self.per = per
        self.extent = api.pivot_extent(piv_off, piv_uv) if self.n else (1, 0, 0)
        self.d_xy = torch.from_numpy(np.ascontiguousarray(xy)).to(dev)
        self.d_uv = torch.from_numpy(np.ascontiguousarray(piv_uv)).to(dev)
        self.d_off = torch.from_numpy(np.ascontiguousarray(piv_off)).to(dev)
        self.d_out = torch.full((per, 3), float("nan"), dtype=torch.float32, device=dev)


def timed(torch, dist, world, dev, step, steps, warmup):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
    Returns (seconds, mean ms between the HIP events recorded around the kernel launch of each step, per-rank record):
    at N > 1 the record holds every rank's own kernel time and the time of its exchange (a third event behind the all-gather),
    so that a scaling line shows which rank and which part set the step."""
    for _ in range(warmup):
        step(None)
    events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(events[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    per_rank = None
    if world > 1:
        gather_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
        mine = torch.tensor([kern_ms, gather_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = {"kernel_ms": [float(t[0].item()) for t in every], "all_gather_ms": [float(t[1].item()) for t in every],
                    "what": "per rank, mean over the timed steps: HIP events around the matcher launch, and from there to behind the "
                            "all-gather (which waits for the slowest rank's kernel)"}
    return elapsed, kern_ms, per_rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C4"])
    ap.add_argument("--path", default="auto", choices=["auto", "u8px", "general", "f32", "u16"],
                    help="auto: exact u8 kernel when the pair is 8-bit integral; f32: the register-tiled f32 kernel "
                         "(what 16-bit / filtered imagery gets); general: force the fallback f32 kernel")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="which figure is the headline `value` at N>1 (the other one is reported too): strong = the config's "
                         "points sharded across the ranks (BASELINE configs[2], default); weak = every rank matches its own "
                         "lattice of the config's size")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default). gloo + MIMC3_BENCH_ONE_DEVICE=1 rehearses the N>1 path on a 1-GPU box")
    ap.add_argument("--qm-sweeps", type=int, default=10,
                    help="also time the QM pseudo-smoothing update (BASELINE configs[4]: 10 sweeps fused on device); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-program", action="store_true",
                    help="skip the extra `program` object (the reference program's whole data path, mimc3_vmap, once)")
    ap.add_argument("--no-f32-path", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (register-tiled kernel alone, null-free pair, the program's 32 passes)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="grid points for the CPU baseline (0 = auto)")
    args = ap.parse_args()

    import torch
    from mimc3_amd import api, shard, synth   # raises if libmimc3_hip.so is missing: no CPU fallback

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("MIMC3_BENCH_ONE_DEVICE") == "1":
        local_rank = 0                         # rehearsal only: every rank shares GPU 0 (gloo backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    collective = "none" if world == 1 else ("RCCL all-gather (torch.distributed nccl backend)" if args.backend == "nccl"
                                            else "gloo all-gather through host memory (rehearsal backend)")

    # ---- inputs (seeded; identical images on every rank) -------------------------------------
    case = synth.make_case(args.config)
    H, W = case.i0.shape
    xy_all = case.xyuvav
    n_all = xy_all.shape[0]
    ctx = api.Context(local_rank)
    raw_ok = bool(case.i0.max() <= 255 and case.i1.max() <= 255)
    t0 = time.perf_counter()
    ctx.set_images(case.i0, case.i1)                       # f32 pair from pageable memory (through the pinned staging chunks)
    t_h2d_f32 = time.perf_counter() - t0
    t_h2d_raw = None
    if raw_ok:                                             # the pair as the 8-bit TIFF holds it, from pinned memory
        p0, p1 = api.pinned_empty((H, W), np.uint8), api.pinned_empty((H, W), np.uint8)
        p0[:] = case.i0; p1[:] = case.i1
        t0 = time.perf_counter()
        ctx.set_images_raw(p0, p1)
        t_h2d_raw = time.perf_counter() - t0
    ctx.set_path(args.path)
    stream = torch.cuda.current_stream()

    piv_off, piv_uv = api.get_uv_pivot(xy_all, case.dt, case.mpp, case.ocw, H, W)

    def make_step(leg, gather=None):
        def step(ev):
            if ev is not None:
                ev[0].record(stream)
            if leg.n:
                ctx.matching_ncc_dlc_2_dev(leg.d_xy.data_ptr(), leg.n, case.offset, leg.d_uv.data_ptr(), leg.d_off.data_ptr(),
                                           leg.extent, case.ocw, leg.d_out.data_ptr(), stream=stream.cuda_stream)
            if ev is not None:
                ev[1].record(stream)
            if gather is not None:
                gather(leg)
                if ev is not None:
                    ev[2].record(stream)
        return step

    result = {}
    if world == 1:
        leg = Leg(torch, api, dev, xy_all, piv_off, piv_uv, n_all)
        elapsed, kern_ms, _ = timed(torch, dist, world, dev, make_step(leg), args.steps, args.warmup)
        alg_bytes = algorithmic_bytes(piv_off, piv_uv, case.ocw)
        n_job, n_rank0, scaling = n_all, n_all, "n/a"        # one rank: nothing scales
        got = leg.d_out.cpu().numpy()
        path = ctx.last_path()
    else:
        # ---- strong: BASELINE configs[2], the config's points in cost-balanced shares
        cost = api.point_cost(piv_off, case.ocw)
        order, start, per, imb = shard.balanced_shares(cost, world)
        mine = order[start[rank]:start[rank + 1]]
        sxy, soff, suv = shard.gather_problem(xy_all, piv_off, piv_uv, mine)
        leg_s = Leg(torch, api, dev, sxy, soff, suv, per)
        field = {}
        to_grid = shard.Unpermute(order, start, per, dev)

        def gather_strong(leg):
            g = shard.all_gather_blocks(leg.d_out, per, world)                 # the ONE exchange: [world][per][3]
            field["full"] = to_grid(g)                                         # grid order, on every rank

        el_s, km_s, pr_s = timed(torch, dist, world, dev, make_step(leg_s, gather_strong), args.steps, args.warmup)
        # ---- weak: every rank its own lattice of the config's size (lattice r shifted r px in x: an N-times denser grid)
        xy_w = xy_all.copy()
        xy_w[:, 2] += rank; xy_w[:, 0] += rank * case.mpp
        woff, wuv = api.get_uv_pivot(xy_w, case.dt, case.mpp, case.ocw, H, W)
        leg_w = Leg(torch, api, dev, xy_w, woff, wuv, n_all)

        def gather_weak(leg):
            field["weak"] = shard.all_gather_blocks(leg.d_out, n_all, world)

        el_w, km_w, pr_w = timed(torch, dist, world, dev, make_step(leg_w, gather_weak), args.steps, args.warmup)
        strong = {"value": n_all * args.steps / el_s, "ms_per_step": el_s / args.steps * 1e3, "kernel_ms_rank0": km_s,
                  "grid_points_rank0": int(leg_s.n), "points_per_step": n_all, "work_imbalance": imb, "per_rank": pr_s,
                  "what": f"{args.config}'s {n_all} points in cost-balanced shares of {shard.default_block(n_all, world)}-point blocks"}
        weak = {"value": n_all * world * args.steps / el_w, "ms_per_step": el_w / args.steps * 1e3, "kernel_ms_rank0": km_w,
                "grid_points_rank0": n_all, "points_per_step": n_all * world, "per_rank": pr_w,
                "what": f"every rank its own {n_all}-point lattice (shifted r px), all-gather of [{world}][{n_all}][3]"}
        result["strong"], result["weak"] = strong, weak
        if args.scaling == "strong":
            elapsed, kern_ms, n_job, n_rank0, scaling = el_s, km_s, n_all, leg_s.n, "strong"
            alg_bytes = algorithmic_bytes(soff, suv, case.ocw) if leg_s.n else 0
        else:
            elapsed, kern_ms, n_job, n_rank0, scaling = el_w, km_w, n_all * world, n_all, "weak"
            alg_bytes = algorithmic_bytes(woff, wuv, case.ocw)
        got = field["full"].cpu().numpy()
        path = ctx.last_path()

    if rank == 0:
        value = n_job * args.steps / elapsed
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        kname = KERNEL_NAMES.get(path, path)
        res = {
            "metric": "grid-points/s (DLC NCC match)", "value": value, "unit": "grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None,
            "dtype": DTYPES.get(path, "f32 products, f64 sums"),
            "kernel_path": path,
            "data": "synthetic",
            "config": {"workload": f"{'C3' if (world > 1 and scaling == 'strong' and args.config == 'C2') else args.config}: {W}x{H} synthetic "
                                   f"shifted pair, {n_job} grid points per step ({n_rank0} on rank 0; grid {case.dimx}x{case.dimy}), "
                                   f"ocw {case.ocw} ({2 * case.ocw + 1}^2 chip), up to {int((piv_off[1:] - piv_off[:-1]).max())} pivots, "
                                   f"window up to {2 * (int(np.abs(piv_uv).max()) + case.ocw + 2) + 1}^2",
                       "grid_points_rank0": int(n_rank0), "image": [H, W], "ocw": case.ocw,
                       "parallelism": f"grid-point shard x{world}", "collective": collective},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kname,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "timer": "HIP events recorded on the launch stream around each of the K timed launches (rank 0)"},
            "h2d_images_s": t_h2d_raw if t_h2d_raw is not None else t_h2d_f32,
            "h2d_images": {"raw_dn_pinned_s": t_h2d_raw, "f32_pageable_s": t_h2d_f32,
                           "note": "pair upload incl. device-side widening + plane build; raw = 1 B/px as the 8-bit TIFF holds it"},
        }
        if path == "u8_mfma":
            res["roofline"]["launches"] = ("two launches per step, timed together: match_ncc_dlc_mx (the points without null pixels in reach whose corridors fit its "
                                           "32 x 32-cell tile; it flags the others), then match_ncc_dlc_px<PxU8> in flag mode for the flagged points")
        res.update(result)
        try:   # counters of this kernel from rocprofv3 PMC passes of this same command (tools/profile.sh + tools/pmc_to_json.py)
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))[args.config][kname]
            rf = res["roofline"]
            rf["traffic"] = tr["bytes"]
            rf["traffic_source"] = tr["source"] + " (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; measured in separate --pmc passes, not in this run)"
            # SURVEY 8(d)'s secondary bounds: the kernel is VALU-issue-bound, not HBM-bound (traffic << algorithmic bytes)
            for k in ("valu_per_point", "salu_per_point", "lds_per_point", "valu_busy", "waves_per_simd", "wait_share_of_wave_time",
                      "issue_stall_share_of_wave_time", "issuing_share_of_wave_time", "kernel_ms_profile_avg", "kernels"):
                if k in tr:
                    rf[k] = tr[k]
            # what bounds the pass, by the counters: its HBM traffic is a few percent of the algorithmic bytes, its time goes into issuing
            # (VALU + MFMA + SALU) and into the latency of short dependent chains
            if "bound" in tr:
                rf["bound"] = tr["bound"]
            if "compute" in tr:
                rf["compute"] = tr["compute"]
            if "lds_per_point" in tr and kern_ms > 0:
                # LDS wave-instructions x 64 lanes x 4 B (the kernel's LDS traffic is ds_read_b32 / ds_read2_b32) over the live kernel time;
                # the ds_read_b32 peak of the chip is ~75 TB/s (MI355X_MICROARCH.md, LDS)
                rf["lds_bytes_per_s"] = tr["lds_per_point"] * 256.0 * n_rank0 / (kern_ms * 1e-3)
                rf["lds_peak_bytes_per_s"] = 75e12
            rf["pmc_kernel_sha16"] = tr.get("kernel_sha16")
            rf["pmc_stale"] = tr.get("kernel_sha16") != kernel_source_sha16()     # True: the kernel source changed after the PMC passes
        except Exception as e:      # (no PMC record for this kernel / config: say so instead of dropping the fields silently)
            res["roofline"]["traffic_error"] = f"{type(e).__name__}: {e}"
        valid = got[:, 2] > -2.5
        res["check"] = {"valid_frac": float(valid.mean()),
                        "median_du_dv": [float(np.nanmedian(got[:, 0])), float(np.nanmedian(got[:, 1]))],
                        "true_shift": list(case.shift)}
        if world == 1:
            res["incl_io"] = io_leg(api, ctx, case, xy_all, piv_off, piv_uv, min(args.steps, 10), got)
            res["value_incl_io"] = res["incl_io"]["value"]
            cpu = None
            if not args.no_cpu_baseline:
                res["cpu_baseline"], res["parity"], cpu = cpu_baseline(case, xy_all, piv_off, piv_uv, got, args.cpu_sample)
            if not args.no_f32_path and args.path == "auto" and path != "f32_tiled":
                res["f32_path"] = f32_leg(torch, dist, dev, ctx, leg, make_step, piv_off, piv_uv, case, args, cpu)
                ctx.set_path(args.path)
            if not args.no_legs and args.path == "auto" and path == "u8_mfma":
                res["kernels"] = kernel_legs(torch, dist, dev, api, synth, ctx, leg, make_step, piv_off, piv_uv, case, args, got)
                ctx.set_path(args.path)
        if args.qm_sweeps > 0:
            res["qm"] = qm_leg(torch, api, synth, ctx, dev, case, args.qm_sweeps, check=(world == 1 and not args.no_cpu_baseline))
        if world == 1 and not args.no_program:
            res["program"] = program_leg(api, ctx, xy_all)
            if not args.no_legs:
                res["program"]["passes"] = passes_leg(torch, api, ctx, dev, case, xy_all)
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def io_leg(api, ctx, case, xy, piv_off, piv_uv, steps, want):
    """one pass INCLUDING everything the host hands over and takes back per pass; the images stay resident
    as in the CLI's 32-pass schedule.  Two forms, both from pinned host arrays:
      geo  mimc3_match_ncc_dlc_geo: get_uv_pivot + matching_ncc_dlc_2 in one call -- the host computes the corridor of every
           point (the libm half of get_uv_pivot: INSIDE the timed step), uploads xyuvav + 24 B of corridor per point, the
           pivot lists are expanded on the device;
      csr  mimc3_match_ncc_dlc: the pivot CSR made beforehand on the host is uploaded with xyuvav (round 2's form)."""
    n = xy.shape[0]
    pxy = api.pinned_empty(xy.shape, np.float64); pxy[:] = xy
    puv = api.pinned_empty(piv_uv.shape, np.int32); puv[:] = piv_uv
    poff = api.pinned_empty(piv_off.shape, np.int64); poff[:] = piv_off

    def run(fn):
        # every call is a complete host-to-host pass (synchronous): timed one by one, the MEDIAN of the steps is reported -- a host
        # hiccup in one of ten steps (seen: one 11 ms step among 3 ms ones) would otherwise decide the figure
        fn(); fn()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), out

    cor = api.pivot_corridors(xy, case.dt, case.mpp)                  # the libm half of get_uv_pivot: made beforehand, like the lists of the csr form
    pcor = api.pinned_empty(cor.shape, np.uint8); pcor[:] = cor
    pout = api.pinned_empty((n, 3), np.float32)
    dt_cor, got = run(lambda: ctx.matching_ncc_dlc_cor(pxy, pcor, case.offset, case.ocw, out=pout))
    got = np.array(got)
    pout2 = api.pinned_empty((n, 3), np.float32)
    dt_geo, got_geo = run(lambda: ctx.matching_ncc_dlc_geo(pxy, case.offset, case.dt, case.mpp, case.ocw, out=pout2))
    same_geo = bool(np.array_equal(np.asarray(got_geo).view(np.uint32), got.view(np.uint32)))
    dt_csr, _ = run(lambda: ctx.matching_ncc_dlc_2(pxy, case.offset, poff, puv, case.ocw))
    same = bool(np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(want).view(np.uint32)))
    return {"value": n / dt_cor, "ms_per_step": dt_cor * 1e3, "steps": steps, "statistic": "median of the steps (each a synchronous host-to-host call)",
            "bytes_over_pcie_per_step": int((16 + api.CORRIDOR_BYTES + 12) * n + 4 * 24),
            "identical_to_resident_run": same,
            "what": "mimc3_match_ncc_dlc_cor: per grid point 16 B of (u, v) + 24 B of corridor up, pivot lists made on the device, kernel, 12 B down; "
                    "three chunks, transfers under the matcher; pair resident, corridors made beforehand (as the csr form's lists are)",
            "with_host_corridors_inside": {"value": n / dt_geo, "ms_per_step": dt_geo * 1e3, "identical": same_geo,
                                           "what": "mimc3_match_ncc_dlc_geo: + atan2 / cos / sin of every point on the host threads inside the step"},
            "csr_upload_form": {"value": n / dt_csr, "ms_per_step": dt_csr * 1e3,
                                "bytes_over_pcie_per_step": int(pxy.nbytes + puv.nbytes + poff.nbytes + 12 * n),
                                "what": "mimc3_match_ncc_dlc: xyuvav + host-made pivot CSR up (round 2's form)"}}


def f32_leg(torch, dist, dev, ctx, leg, make_step, piv_off, piv_uv, case, args, cpu):
    """The same C2 workload forced onto the register-tiled f32 kernel (f32 products, f64 sums): what a 16-bit pair gets."""
    ctx.set_path("f32")
    leg.d_out.fill_(float("nan"))
    elapsed, kern_ms, _ = timed(torch, dist, 1, dev, make_step(leg), args.steps, 2)
    alg = algorithmic_bytes(piv_off, piv_uv, case.ocw)
    ach = alg / (kern_ms * 1e-3) / 1e9
    out = {"kernel_path": ctx.last_path(), "value": leg.n * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "kernel": KERNEL_NAMES.get(ctx.last_path(), ctx.last_path()), "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg}}
    if cpu is not None:
        idx, ref = cpu
        g = leg.d_out.cpu().numpy()[idx]
        nan_same = bool(np.array_equal(np.isnan(g), np.isnan(ref)))
        out["parity"] = {"points": int(len(idx)), "invalid_mask_equal": nan_same,
                         "max_abs_diff_px": float(np.nanmax(np.abs(g - ref))) if np.isfinite(ref).any() else 0.0,
                         "bit_identical": bool(nan_same and np.array_equal(np.nan_to_num(g).view(np.uint32), np.nan_to_num(ref).view(np.uint32)))}
    return out


def kernel_legs(torch, dist, dev, api, synth, ctx, leg, make_step, piv_off, piv_uv, case, args, got):
    """Secondary figures beside `value` (same workload, same timer): the register-tiled kernel alone (mode "u8px": what round 3 measured),
    and the same grid on a NULL-FREE pair -- every point then takes the matrix-core kernel's clean form, which is the rate of imagery
    without void areas (the config's pair has 2 % of each image zeroed in blobs: 52 % of its points hold a null in chip or window and
    are the register-tiled kernel's)."""
    out = {}
    ctx.set_path("u8px")
    leg.d_out.fill_(float("nan"))
    el, km, _ = timed(torch, dist, 1, dev, make_step(leg), args.steps, 2)
    g = leg.d_out.cpu().numpy()
    out["register_tiled_alone"] = {"kernel_path": ctx.last_path(), "ms_per_step": el / args.steps * 1e3, "kernel_ms": km, "value": leg.n * args.steps / el,
                                   "bit_identical_to_headline": bool(np.array_equal(np.isnan(g), np.isnan(got)) and
                                                                      np.array_equal(np.nan_to_num(g).view(np.uint32), np.nan_to_num(got).view(np.uint32)))}
    clean = synth.make_case(args.config, null_frac=0.0)
    ctx.set_images(clean.i0, clean.i1)
    res = {}
    for mode in ("auto", "u8px"):
        ctx.set_path(mode)
        leg.d_out.fill_(float("nan"))
        el, km, _ = timed(torch, dist, 1, dev, make_step(leg), args.steps, 2)
        res[mode] = (el, km, ctx.last_path(), leg.d_out.cpu().numpy())
    same = bool(np.array_equal(np.nan_to_num(res["auto"][3]).view(np.uint32), np.nan_to_num(res["u8px"][3]).view(np.uint32)))
    out["null_free_pair"] = {"what": "the same grid on the pair without the zeroed blobs: every point on the matrix-core kernel's clean form",
                             "ms_per_step": res["auto"][0] / args.steps * 1e3, "kernel_ms": res["auto"][1], "value": leg.n * args.steps / res["auto"][0],
                             "kernel_path": res["auto"][2], "register_tiled_ms_per_step": res["u8px"][0] / args.steps * 1e3,
                             "bit_identical_to_register_tiled": same}
    ctx.set_images(case.i0, case.i1)
    return out


def passes_leg(torch, api, ctx, dev, case, xy):
    """The 32 matcher passes of the program (MIMC_main.c:261-350: chip sizes 7 / 15 / 30 / 40 x raw, d/dx, d/dy, Laplacian x forward,
    swapped), each timed on its own with HIP events on the resident pair: kernel path, ms, algorithmic bytes and the
    fraction of the HBM roofline they amount to -- the program's seconds have a roofline of their own."""
    H, W = case.i0.shape
    stream = torch.cuda.current_stream()
    d_xy = torch.from_numpy(np.ascontiguousarray(xy)).to(dev)
    d_out = torch.empty((xy.shape[0], 3), dtype=torch.float32, device=dev)
    piv = {}
    for ocw in (7, 15, 30, 40):
        off, uv = api.get_uv_pivot(xy, case.dt, case.mpp, ocw, H, W)
        piv[ocw] = (off, uv, api.pivot_extent(off, uv), torch.from_numpy(off).to(dev), torch.from_numpy(uv).to(dev), torch.from_numpy(np.ascontiguousarray(-uv)).to(dev),
                    algorithmic_bytes(off, uv, ocw))
    ctx.set_path("auto")
    ctx.enable_timing(True)
    out = []
    for vi, variant in enumerate(("raw", "ddx", "ddy", "laplacian")):
        ctx.filter_images(None if vi == 0 else api.CLI_KERNELS[vi - 1])
        for ocw in (7, 15, 30, 40):
            off, uv, ext, d_off, d_uv, d_uvn, alg = piv[ocw]
            for swap in (False, True):
                best = None
                for _ in range(2):
                    ctx.matching_ncc_dlc_2_dev(d_xy.data_ptr(), xy.shape[0], -case.offset if swap else case.offset, (d_uvn if swap else d_uv).data_ptr(),
                                               d_off.data_ptr(), ext, ocw, d_out.data_ptr(), swap=swap, stream=stream.cuda_stream)
                    torch.cuda.synchronize()
                    ms = ctx.last_kernel_ms()
                    best = ms if best is None else min(best, ms)
                ach = alg / (best * 1e-3) / 1e9
                out.append({"variant": variant, "ocw": ocw, "direction": "swapped" if swap else "forward", "kernel_path": ctx.last_path(),
                            "kernel": KERNEL_NAMES.get(ctx.last_path(), ctx.last_path()), "ms": best, "algorithmic_bytes": alg,
                            "frac": ach / HBM_PEAK_GBS})
    ctx.filter_images(None)
    ctx.enable_timing(False)
    return out


def program_leg(api, ctx, xy):
    """Informational, not the metric: the reference program's whole data path (MIMC_main.c:203-402 = CP offset, 32
    matcher passes on the raw and the three filtered pairs, clustering, dpf0/dpf1, QM, unit conversion) as ONE call on
    the resident pair.  5 % of the grid is given a slow a-priori so that the control-point stage has candidates."""
    x = np.array(xy, np.float64, copy=True)
    rng = np.random.default_rng(1)
    slow = rng.random(x.shape[0]) < 0.05
    x[slow, 4] = rng.uniform(-5, 5, slow.sum()); x[slow, 5] = rng.uniform(-5, 5, slow.sum())
    best = None
    for _ in range(2):
        t = time.time()
        out = ctx.vmap(x, 16.0, cp_seed=7)
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
    ok = out["cp_status"] > 0
    return {"what": "mimc3_vmap: CP offset + 32 matcher passes (ocw 7/15/30/40 x raw/ddx/ddy/laplacian x fwd/swapped) + postprocess",
            "seconds": best, "grid_points": int(x.shape[0]), "cp_status": int(out["cp_status"]), "cp_offset": list(out["offset_cp"]),
            "finite_frac": float(np.isfinite(out["vx"]).mean()) if ok else None,
            "reference": "the unmodified reference program on the same inputs is timed by tests/fullsize/vmap_fullsize.py "
                         "(profiles/round*/vmap_fullsize_*.json); not re-timed here (minutes of CPU)"}


def qm_leg(torch, api, synth, ctx, dev, case, sweeps, check, reps=5):
    """Secondary figure (does not enter `value`): `sweeps` QM pseudo-smoothing sweeps on the config's
    grid, all enqueued on the device without host synchronisation (get_dpf_pseudosmoothing,
    MIMC_module.c:1986-2312), on synthetic candidate clusters; checked against the oracle."""
    dimx, dimy = case.dimx, case.dimy
    mvn, nclus, dpf, dx, dy = synth.synth_qm_state(dimx, dimy, seed=20260105)
    xyg = case.xyuvav
    mps = float(np.float32(xyg[1, 0] - xyg[0, 0]))
    ruv = api.get_ruv_neighbor(xyg, dimx, dimy, mps, 5.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_mvn, d_ncl, d_xy, d_ruv = t(mvn), t(nclus), t(xyg), t(ruv)
    src = (t(dpf), t(dx), t(dy))
    wrk = (torch.empty_like(src[0]), torch.empty_like(src[1]), torch.empty_like(src[2]))
    d_work = torch.empty(ctx.qm_workspace_bytes(dimx * dimy, sweeps), dtype=torch.uint8, device=dev)
    d_sw = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    times = []
    for _ in range(reps + 1):
        for a, b in zip(wrk, src):
            a.copy_(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        ctx.get_dpf_pseudosmoothing_dev(dimy, dimx, wrk[0].data_ptr(), wrk[1].data_ptr(), wrk[2].data_ptr(), d_ruv.data_ptr(),
                                        ruv.shape[0], d_mvn.data_ptr(), mvn.shape[1], d_ncl.data_ptr(), d_xy.data_ptr(), sweeps,
                                        d_work.data_ptr(), d_sw.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = float(np.median(times[1:]))
    nsw = int(d_sw.item())
    out = {"sweeps_cap": sweeps, "sweeps_run": nsw, "grid": [dimy, dimx], "neighbours": int(ruv.shape[0]),
           "ms": ms, "grid_points_per_s": dimx * dimy / (ms * 1e-3),
           "sweeps_per_s": max(nsw, 1) / (ms * 1e-3), "launches_per_sweep": api.qm_launches_per_sweep(),
           "launches_enqueued": 1 + sweeps * api.qm_launches_per_sweep(),
           "bound": "latency: per sweep one f64 fit kernel (<= 81 neighbour gathers + 6x6 Gauss-Jordan per investigated point, "
                    "thread per point) + one commit/compare/decide launch; the data (~1.4 KB per investigated point) is L2-resident",
           "investigated_frac_initial": float((mvn[np.arange(dimx * dimy), dpf.reshape(-1), 4] < 0.6).mean())}
    if check:
        from oracle import oracle as orc
        o = orc.Oracle("port")
        t0 = time.perf_counter()
        wd, wx, wy, st = o.qm(dpf, dx, dy, ruv, mvn, nclus, xyg, max_sweeps=sweeps)
        out["cpu_ms"] = (time.perf_counter() - t0) * 1e3
        out["cpu_kind"] = "port (the reference's QM is serial too)"
        gd, gx, gy = wrk[0].cpu().numpy(), wrk[1].cpu().numpy(), wrk[2].cpu().numpy()
        out["identical"] = bool(np.array_equal(gd, wd) and np.array_equal(np.nan_to_num(gx).view(np.uint32), np.nan_to_num(wx).view(np.uint32))
                                and np.array_equal(np.nan_to_num(gy).view(np.uint32), np.nan_to_num(wy).view(np.uint32)))
        out["sweeps_cpu"] = int(st[0])
        out["points_changed"] = int((gd != dpf).sum())
    return out


def cpu_baseline(case, xy, piv_off, piv_uv, gpu_out, sample):
    """Time the CPU path on this box's host cores on a bounded sample of the SAME workload, and
    use its output as the parity checker for the GPU result (the oracle is only the checker)."""
    from oracle import oracle as orc
    kind = "reference" if orc.available("reference") else "port"
    if kind == "port" and not orc.available("port"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    o = orc.Oracle(kind)
    n = xy.shape[0]
    cores = o.num_threads()
    if sample <= 0:
        # aim at ~15 s of CPU work: calibrate on 2,000 points
        idx = np.arange(0, n, max(1, n // 2000))[:2000]
        t0 = time.perf_counter()
        _subset_match(o, case, xy, piv_off, piv_uv, idx)
        rate = len(idx) / (time.perf_counter() - t0)
        sample = int(min(n, max(2000, rate * 15.0)))
    idx = np.unique(np.linspace(0, n - 1, sample).astype(np.int64))
    t0 = time.perf_counter()
    cpu = _subset_match(o, case, xy, piv_off, piv_uv, idx)
    dt = time.perf_counter() - t0
    g = gpu_out[idx]
    nan_same = bool(np.array_equal(np.isnan(g), np.isnan(cpu)))
    diff = float(np.nanmax(np.abs(g - cpu))) if np.isfinite(cpu).any() else 0.0
    bits = bool(nan_same and np.array_equal(np.nan_to_num(g).view(np.uint32), np.nan_to_num(cpu).view(np.uint32)))
    base = {"value": len(idx) / dt, "unit": "grid-points/s", "cores": cores, "kind": kind,
            "sample": f"{len(idx)} of {n} grid points (evenly spaced), one matcher pass, {dt:.1f} s, OpenMP dynamic"}
    par = {"points": int(len(idx)), "invalid_mask_equal": nan_same, "max_abs_diff_px": diff, "bit_identical": bits}
    return base, par, (idx, cpu)


def _subset_match(o, case, xy, piv_off, piv_uv, idx):
    from mimc3_amd import shard
    sxy, off, uv = shard.gather_problem(xy, piv_off, piv_uv, idx)
    return o.match(case.i0, case.i1, sxy, case.offset, off, uv, case.ocw)


if __name__ == "__main__":
    main()
