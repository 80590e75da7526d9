#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: grid-points/s of the DLC/NCC matcher.

    python bench.py --gpus N --steps K --warmup W           (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one matcher pass (matching_ncc_dlc_2) over one batch of synthetic grid points.
Workload at N=1 = BASELINE configs[1]: 4096x4096 synthetic pair, 200,000 grid points (500x400),
ocw 16 (33x33 chip), ~15 DLC pivots (65x65 window).  Inputs (both images, xyuvav, pivot CSR) are
resident in HBM before the timed region.  At N>1 every rank matches its own 200,000-point lattice
on the replicated pair (lattice r is shifted r px in x: together an N-times denser grid; weak
scaling), then the (u,v,ncc) field is re-assembled on every GPU with one RCCL all-gather.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and, at N=1,
`cpu_baseline` (the compiled reference OpenMP path when oracle/_ref travelled here, else the
parity-verified C restatement) and a parity figure against it.
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


def algorithmic_bytes(piv_off, piv_uv, ocw):
    """B = 4(2ocw+1)^2 + 4*Dy2*Dx2 + 48 + 8*npiv + 12 per grid point, summed."""
    npiv = (piv_off[1:] - piv_off[:-1]).astype(np.int64)
    last = piv_uv[piv_off[1:] - 1].astype(np.int64)
    dx2 = np.abs(last[:, 0]) + ocw + 2
    dy2 = np.abs(last[:, 1]) + ocw + 2
    chip = 4 * (2 * ocw + 1) ** 2
    return int((chip + 4 * (2 * dx2 + 1) * (2 * dy2 + 1) + 48 + 8 * npiv + 12).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C4"])
    ap.add_argument("--path", default="auto", choices=["auto", "general", "f32", "u16"],
                    help="auto: exact u8 kernel when the pair is 8-bit integral; f32: the register-tiled f32 kernel "
                         "(what 16-bit / filtered imagery gets); general: force the fallback f32 kernel")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank matches its own lattice of the config's size (default); "
                         "strong: the config's points are sharded across ranks (BASELINE configs[2])")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default). gloo + MIMC3_BENCH_ONE_DEVICE=1 rehearses the N>1 path on a 1-GPU box")
    ap.add_argument("--qm-sweeps", type=int, default=10,
                    help="also time the QM pseudo-smoothing update (BASELINE configs[4]: 10 sweeps fused on device); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-program", action="store_true",
                    help="skip the extra `program` object (the reference program's whole data path, mimc3_vmap, once)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="grid points for the CPU baseline (0 = auto)")
    args = ap.parse_args()

    import torch
    from mimc3_amd import api, synth   # raises if libmimc3_hip.so is missing: no CPU fallback

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
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

    # ---- inputs (seeded; identical images on every rank) -------------------------------------
    case = synth.make_case(args.config)
    H, W = case.i0.shape
    from mimc3_amd import shard
    xy = case.xyuvav.copy()
    n_job = xy.shape[0] * (world if args.scaling == "weak" else 1)   # grid points of the whole job per step
    if args.scaling == "weak":
        if rank:                               # rank r's lattice: shifted r px in x
            xy[:, 2] += rank
            xy[:, 0] += rank * case.mpp
        per = xy.shape[0]
    else:
        lo, hi, per = shard.block_range(xy.shape[0], world, rank)
        xy = np.ascontiguousarray(xy[lo:hi])
    n = xy.shape[0]
    piv_off, piv_uv = api.get_uv_pivot(xy, case.dt, case.mpp, case.ocw, H, W)
    extent = api.pivot_extent(piv_off, piv_uv)
    alg_bytes = algorithmic_bytes(piv_off, piv_uv, case.ocw)

    t_h2d0 = time.perf_counter()
    d_i0 = torch.from_numpy(case.i0).to(dev)
    d_i1 = torch.from_numpy(case.i1).to(dev)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t_h2d0
    d_xy = torch.from_numpy(xy).to(dev)
    d_uv = torch.from_numpy(piv_uv).to(dev)
    d_off = torch.from_numpy(piv_off).to(dev)
    d_out = torch.full((per, 3), float("nan"), dtype=torch.float32, device=dev)   # padded to the block size
    d_all = torch.empty((world * per, 3), dtype=torch.float32, device=dev) if world > 1 else None

    ctx = api.Context(local_rank)
    t_prep0 = time.perf_counter()
    ctx.set_images_dev(d_i0.data_ptr(), d_i1.data_ptr(), H, W, keep=(d_i0, d_i1))   # builds + proves the u8 planes
    t_prep = time.perf_counter() - t_prep0
    ctx.set_path(args.path)
    stream = torch.cuda.current_stream()

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        ctx.matching_ncc_dlc_2_dev(d_xy.data_ptr(), n, case.offset, d_uv.data_ptr(), d_off.data_ptr(), extent,
                                   case.ocw, d_out.data_ptr(), stream=stream.cuda_stream)
        if ev is not None:
            ev[1].record(stream)
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(d_all, d_out)      # RCCL over xGMI: the one exchange step
            else:
                dist.all_gather(list(d_all.view(world, per, 3).unbind(0)), d_out)

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    if rank == 0:
        total_pts = n_job * args.steps
        value = total_pts / elapsed
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "grid-points/s (DLC NCC match)", "value": value, "unit": "grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": ("u8 pixels, exact u32 dot4 sums, f64 NCC" if ctx.last_path() == "u8_exact"
                      else "f32 pixels, f32 products, f64 sums and NCC"),
            "kernel_path": ctx.last_path(),
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H} synthetic shifted pair, {n} grid points on rank 0 of {n_job} per step "
                                   f"({case.dimx}x{case.dimy}), ocw {case.ocw} ({2 * case.ocw + 1}^2 chip), "
                                   f"{extent[0]} pivots max, window up to {2 * (extent[1] + case.ocw + 2) + 1}^2",
                       "grid_points_per_gpu": n, "image": [H, W], "ocw": case.ocw,
                       "parallelism": f"grid-point shard x{world}" + (", RCCL all-gather of [N,3]" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": {"u8_exact": "match_ncc_dlc_u8", "f32_tiled": "match_ncc_dlc_px<PxF32>", "u16_scaled": "match_ncc_dlc_px<PxU16>"}.get(ctx.last_path(), "match_ncc_dlc_f32"),
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "h2d_images_s": t_h2d, "u8_plane_prep_s": t_prep,
        }
        try:   # HBM bytes per launch measured by rocprofv3 PMC passes of this same command (profiles/)
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))[args.config][res["roofline"]["kernel"]]
            res["roofline"]["traffic"] = tr["bytes"]
            res["roofline"]["traffic_source"] = tr["source"] + " (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"
        except Exception:
            pass
        got = d_out[:n].cpu().numpy()
        valid = got[:, 2] > -2.5
        res["check"] = {"valid_frac": float(valid.mean()),
                        "median_du_dv": [float(np.nanmedian(got[:, 0])), float(np.nanmedian(got[:, 1]))],
                        "true_shift": list(case.shift)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], res["parity"] = cpu_baseline(case, xy, piv_off, piv_uv, got, args.cpu_sample)
        if args.qm_sweeps > 0:
            res["qm"] = qm_leg(torch, api, synth, ctx, dev, case, args.qm_sweeps, check=(world == 1 and not args.no_cpu_baseline))
        if world == 1 and not args.no_program:
            res["program"] = program_leg(api, ctx, xy)
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


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
            "reference_program_seconds_same_box": 323.6, "reference_source": "profiles/round1/vmap_fullsize_C2.json (256 host threads)"}


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
    out = {"sweeps_cap": sweeps, "sweeps_run": int(d_sw.item()), "grid": [dimy, dimx], "neighbours": int(ruv.shape[0]),
           "ms": ms, "grid_points_per_s": dimx * dimy / (ms * 1e-3),
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
    return base, par


def _subset_match(o, case, xy, piv_off, piv_uv, idx):
    cnt = (piv_off[idx + 1] - piv_off[idx]).astype(np.int64)
    off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    sel = np.concatenate([np.arange(piv_off[i], piv_off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
    return o.match(case.i0, case.i1, xy[idx], case.offset, off, piv_uv[sel], case.ocw)


if __name__ == "__main__":
    main()
