#!/usr/bin/env python3
"""Benchmark of the graph-Laplacian image filter hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

One step = one whole pass of the approximate path (affinity/degree -> L_A ->
inverse subspace iteration -> Nystroem contraction -> spectral filter) over one
4096x4096 synthetic noisy image at 0.5 % sampling, m = 64 eigenpairs, with the
image already resident in HBM and the filtered image left in HBM. With N > 1 the
SAME image is split by pixel rows over the N ranks (strong scaling); the driver
launches one rank per GPU through torch.distributed.run and the all-reduces
(degree vector, Phi^T y) go over RCCL.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "image-processing-graph-laplacian_amd"))


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU
    through torch.distributed.run, as the driver does) BEFORE this process touches a GPU, and leave with the ranks' exit
    code. Fewer than N visible devices is an error, never a silent one-rank run (the reference's program IS the multi-rank
    program: hpc/image_processing.c:30-76)."""
    if "WORLD_SIZE" in os.environ or "--loopback" in sys.argv:
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args()[0].gpus
    if n <= 1:
        return
    import torch  # (device_count does not initialise the GPU)
    have = torch.cuda.device_count()
    if have < n:
        print(json.dumps({"error": "--gpus %d but only %d device(s) visible" % (n, have), "n_gpus_visible": have}))
        sys.exit(3)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


_self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import glf  # noqa: E402  (HIP path; raises if libglf.so is missing)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)" dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA" dense
PEAK_HBM_GBS = 8000.0
# glf_synth_image(size, size, seed 0): the workload is pinned byte for byte (same table as tests/test_abi.py)
SYNTH_CRC32 = {1024: 0x3d97f4f9, 2048: 0x40fa122b, 4096: 0xdc7203de}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096, help="image is size x size (headline: 4096)")
    ap.add_argument("--sample-frac", type=float, default=0.005)
    ap.add_argument("--num-eigvals", type=int, default=64)
    ap.add_argument("--epsilon", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-skip-leg", action="store_true",
                    help="do not time the extra exact-zero-skipping leg (reported beside the headline)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-nlm-leg", action="store_true",
                    help="skip the non-local-means leg (1024^2, 0.5 % samples, m = 64: the f1 affinity through the whole path)")
    ap.add_argument("--no-batch-leg", action="store_true",
                    help="skip the throughput-mode leg (BASELINE config 5: 64 tiles of 1024x1024 dealt to concurrent contexts)")
    ap.add_argument("--batch-tiles", type=int, default=64)
    ap.add_argument("--batch-tile-size", type=int, default=1024)
    ap.add_argument("--batch-contexts", type=int, default=8)
    ap.add_argument("--force-comm", action="store_true",
                    help="diagnostic: one rank, but with the RCCL collectives in place (cost of the N > 1 plumbing)")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl",
                    help="N > 1 collectives: rccl = issued by the library itself on its stream (glf_ctx_set_comm_rccl, "
                         "default); torch = torch.distributed callbacks through ctypes (the round-1 path, kept for comparison)")
    ap.add_argument("--no-exact-leg", action="store_true",
                    help="skip the exact_grid_form leg (the grid-factored contractions carrying all 256 grey levels: GLF_NYS_PATH / GLF_MV_PATH = grid)")
    ap.add_argument("--no-host-leg", action="store_true",
                    help="skip the host_to_host leg (pinned host image -> pinned host result, copies on the context's stream)")
    ap.add_argument("--loopback", type=int, default=0,
                    help="plumbing rehearsal, NOT a measurement: N ranks of the sharded path time-share ONE GPU through glf.Multi's "
                         "loopback backend (host-staged collectives); prints per-rank stage times and the collectives per step")
    ap.add_argument("--no-direct-leg", action="store_true",
                    help="skip the direct_contraction leg (one step with the entry-by-entry Nystroem kernel, ~0.5 s)")
    return ap.parse_args()


def cpu_baseline(img, info, args):
    """The fp64 CPU oracle (kind 'port': PETSc is not installable) timed on a bounded
    slice of THIS workload and scaled: a band of pixel rows for the two K_B-bound stages, a band
    of L_A rows for the eigen-solver's mat-vecs (iteration counts taken from the GPU run, which
    follows the same algorithm), Gram-Schmidt once. 10-30 s of CPU work in total."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    h, w = img.shape
    N = h * w
    p, m = info["p"], info["m"]
    idx = glf.Sampling(w, h, int(N * args.sample_frac))
    cores = orc.num_threads()
    # affinity/degree: rows band, all p samples
    rows_deg = max(1, min(h, int(round(4e9 / (p * w)))))        # ~4e9 kernel evaluations
    t0 = time.time()
    orc.degree(img, idx, row0=h // 2, row1=h // 2 + rows_deg)
    t_deg = (time.time() - t0) * h / rows_deg
    # L_A rows band + block mat-vec on it (D_A of the GPU run: the diagonal of L_A)
    D = np.ascontiguousarray(info["capture"]["degree"]) if "capture" in info else np.full(p, 1800.0)
    band = max(32, min(p, int(2e9 / p)))                          # ~2e9 entries
    t0 = time.time()
    Arows = orc.laplacian_rows(img, idx, D, info["alpha"], 0, band)
    t_lap = (time.time() - t0) * p / band
    X = orc.random_vectors(p, m, 3)
    t0 = time.time()
    reps = 2
    for _ in range(reps):
        orc.matvec_rows(Arows, X)
    t_mv = (time.time() - t0) / reps * p / band
    del Arows
    t0 = time.time()
    orc.orthonormalise(X)
    t_gs = time.time() - t0
    n_mv = info["inner_its_total"] + info["outer_its"] + 1       # CG mat-vecs + one per residual
    t_eig = n_mv * t_mv + (info["outer_its"] + 1) * t_gs
    # Nystroem: rows band, all samples, m columns
    rows_nys = max(1, min(h, int(round(6e8 / (p * w))) or 1))    # ~6e8 kernel evaluations x m FMAs
    phiA = orc.random_vectors(p, m, 4)
    t0 = time.time()
    orc.nystroem_rows(img, idx, info["alpha"], phiA, np.linspace(0.1, 0.4, m), h // 2, h // 2 + rows_nys)
    t_nys = (time.time() - t0) * h / rows_nys
    total = t_deg + t_lap + t_eig + t_nys
    return {
        "value": round(N / total * 1e-6, 6), "unit": "Mpixel/s", "cores": cores, "kind": "port",
        "sample": ("fp64 oracle: degree on %d of %d pixel rows, L_A build + block mat-vec on %d of %d rows "
                   "(x %d mat-vecs from the GPU run's iteration counts), Gram-Schmidt p x m once (x %d), "
                   "Nystroem on %d of %d pixel rows; each stage scaled to the full image; filter stage "
                   "(N m flops) neglected" % (rows_deg, h, band, p, n_mv, info["outer_its"] + 1, rows_nys, h)),
        "seconds_est": {"affinity": round(t_deg, 1), "laplacian": round(t_lap, 1), "eigen": round(t_eig, 1),
                        "nystroem": round(t_nys, 1), "total": round(total, 1)},
    }


def cpu_parity_cfg2(ctx):
    """Second half of the cpu_baseline leg: the oracle's WHOLE path on BASELINE config 2 (barbara
    512x512, 1 %, m = 64; ~15 s of CPU, the 4096^2 run would take about two hours), timed, and used as the
    checker of the HIP output on the same input: PSNR on the 8-bit images, rel-L2 on the floats."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    from PIL import Image
    img = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "barbara.png")))
    opt = glf.default_options(num_samples=2621, num_eigvals=64, epsilon=0.1)
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    t0 = time.time()
    zf_ref, out_ref, ref = orc.image_processing(img, 2621, 64, epsilon=0.1, inner_rtol=1e-5, seed=1)
    cpu_s = time.time() - t0
    out, zf = out.cpu().numpy(), zf.cpu().numpy()
    mse = float(np.mean((out.astype(np.float64) - out_ref) ** 2))
    return {
        "workload": "barbara.png 512x512, 1% samples (p=2601), m=64, eps=0.1 vs fp64 oracle",
        "cpu_seconds": round(cpu_s, 2), "cpu_mpixel_per_s": round(img.size / cpu_s * 1e-6, 5),
        "gpu_ms": round(info["ms_total"], 3),
        "psnr_db": None if mse == 0 else round(10 * np.log10(255.0 ** 2 / mse), 2),
        "identical_u8": mse == 0,
        "rel_l2_float": float(np.linalg.norm(zf - zf_ref) / np.linalg.norm(zf_ref)),
        "outer_its": [info["outer_its"], ref["outer_its"]],
    }


def parity_headline(ctx, img, cap_run, args):
    """Sampled-row check of the benchmark-size run itself (the oracle's whole path would take ~2 h at 4096^2): the
    oracle's Nystroem rows (hpc/nystroem.c:41-57) from the run's own Phi_A / eigenvalues / alpha and the filter
    (hpc/display.c:58-83) on rows 0, the pass boundary H/2 - 1 | H/2, and H - 1, against the run's Phi, correction and u8
    output; plus D_A on a few samples. tests/test_gpu_large.py asserts the same quantities on more rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    import parity
    out, zf, info = cap_run
    cap = info["capture"]
    size = img.shape[0]
    m = info["m"]
    idx = glf.Sampling(size, size, int(size * size * args.sample_frac))
    t0 = time.time()
    sub = np.unique(np.linspace(0, idx.size - 1, 12).astype(np.int64))
    d_rel = float(np.max(np.abs(cap["degree"][sub] / orc.degree(img, idx[sub]) - 1.0)))
    rows = sorted({0, size // 2 - 1, size // 2, size - 1})
    phi_v = cap["phi"].view(size, size, cap["ld"])
    corr_v = cap["corr"].view(size, size)
    res = parity.check_rows(img, idx, info["alpha"], cap["phi_A"][:, :m].cpu().numpy(), info["eigvals"], cap["c"], rows,
                            phi_gpu=lambda r: phi_v[r, :, :m].cpu().numpy(), zf_gpu=lambda r: zf[r].cpu().numpy(),
                            out_gpu=lambda r: out[r].cpu().numpy(), gain=3.0, corr_gpu=lambda r: corr_v[r].cpu().numpy())
    res["degree_max_rel_err_12_samples"] = d_rel
    res["how_much_the_filter_did"] = ("psnr_ref_vs_input_db = PSNR of the reference's 8-bit output against the INPUT on these rows (inf: equal); "
                                      "rms_correction_grey_levels = RMS of z - y: with the reference's gain 3.0 the filter moves a pixel by ~1e-3 "
                                      "grey levels at this size, so psnr_db / u8_* say little -- tests/test_gpu_large.py has a gain = 2000 case")
    res["cpu_seconds"] = round(time.time() - t0, 1)
    res["what"] = ("rows %s of the %dx%d run vs the fp64 oracle fed the run's Phi_A, eigenvalues, alpha and c = Phi^T y; "
                   "psnr_db / u8_* on the 8-bit output rows, rel_l2_correction on z - y" % (rows, size, size))
    return res


def run_batch_leg(args, ctx, rank, world, barrier):
    """BASELINE config 5: a batch of equally sized tiles (one sample grid), replicas only. The tiles are split evenly over
    the ranks; each rank deals its share to `--batch-contexts` contexts working concurrently (glf_image_processing_batch)
    and, for comparison, runs the same share through one context. No data-path collective; time = max over ranks."""
    ts, total = args.batch_tile_size, args.batch_tiles
    mine = [t for t in range(total) if t % world == rank]
    tiles = np.stack([glf.synth_image(ts, ts, seed=1000 + t) for t in mine]) if mine else np.zeros((0, ts, ts), np.uint8)
    d_tiles = torch.from_numpy(tiles).to(ctx.device)
    opt = glf.default_options(num_samples=int(ts * ts * args.sample_frac), num_eigvals=args.num_eigvals, epsilon=args.epsilon)
    ctxs = [glf.Context(ctx.device.index) for _ in range(max(1, args.batch_contexts))]
    res = {}
    try:
        for label, group in (("one_context", ctxs[:1]), ("concurrent", ctxs)):
            glf.image_processing_batch(group, d_tiles[:min(len(mine), 2 * len(group))].contiguous(), opt)   # warm-up: pools, start block
            barrier()
            t0 = time.perf_counter()
            outs, infos = glf.image_processing_batch(group, d_tiles, opt)
            barrier()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device=ctx.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            res[label] = el
    finally:
        for c in ctxs:
            c.close()
    mp = total * ts * ts * 1e-6
    return {"workload": "%d tiles of %dx%d synthetic, %.1f%% samples (p=%d each), m=%d, eps=%g: every tile through the whole path; "
                        "replicas only (tiles split over %d rank(s), no collective)" % (total, ts, ts, args.sample_frac * 100,
                                                                                        infos[0]["p"] if infos else 0, args.num_eigvals,
                                                                                        args.epsilon, world),
            "contexts_per_gpu": len(ctxs), "value": round(mp / res["concurrent"], 2), "unit": "Mpixel/s",
            "ms_per_tile": round(res["concurrent"] * 1e3 / max(1, total), 3),
            "one_context_value": round(mp / res["one_context"], 2),
            "one_context_ms_per_tile": round(res["one_context"] * 1e3 / max(1, total), 3)}


def run_loopback(args):
    """N ranks on one device: the row-sharded path with every collective in place, staged through the host. What it tells:
    the per-rank compute of an N-way split (stage times of each rank as if it had a GPU to itself are NOT measured -- the
    ranks share the device) and the collectives per step. Flagged as a rehearsal in the line."""
    size, N = args.size, args.size * args.size
    img = glf.synth_image(size, size, seed=0)
    opt = glf.default_options(num_samples=int(N * args.sample_frac), num_eigvals=args.num_eigvals, epsilon=args.epsilon)
    n = args.loopback
    with glf.Multi(n, devices=[0] * n, backend=glf.MULTI_LOOPBACK) as mw:
        for _ in range(args.warmup):
            mw.image_processing(img, opt)
        for r in range(n):
            mw.comm_counters(r, reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, _, infos = mw.image_processing(img, opt)
        el = (time.perf_counter() - t0) / args.steps
        cnt = mw.comm_counters(0, reset=True)
        ci = mw.comm_info(0)
    line = {
        "metric": "filtered Mpixels/sec @ 4K img, 0.5% samples; PSNR vs PETSc ref",
        "measurement": False,
        "what": "plumbing rehearsal: %d ranks of the row-sharded path time-share ONE GPU (glf_multi loopback backend, collectives staged "
                "through host memory, host image in / host image out); value is NOT a multi-GPU number" % n,
        "value": round(N / el * 1e-6, 3), "unit": "Mpixel/s", "n_gpus": 1, "ranks": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el * 1e3, 3), "comm": ci,
        "collectives_per_step_rank0": {k: v / args.steps for k, v in cnt.items()},
        "per_rank": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in inf.items()
                      if k in ("row0", "row1", "ms_affinity", "ms_laplacian", "ms_eigen", "ms_nystroem", "ms_filter", "ms_total", "outer_its",
                               "inner_its_total", "matvecs", "nystroem_path", "matvec_path")} for inf in infos],
        "config": {"workload": "%dx%d synthetic noisy image, %.1f%% samples (p=%d), m=%d eigenpairs, eps=%g" %
                               (size, size, args.sample_frac * 100, infos[0]["p"], infos[0]["m"], args.epsilon)},
    }
    print(json.dumps(line))


def main():
    args = parse_args()
    if args.loopback > 0:
        return run_loopback(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if args.gpus != world:
        if rank == 0:
            print(json.dumps({"error": "--gpus %d but the launcher started WORLD_SIZE %d rank(s)" % (args.gpus, world)}))
        sys.exit(3)

    size = args.size
    N = size * size
    img = glf.synth_image(size, size, seed=0)      # byte-identical on every rank
    import zlib
    crc = zlib.crc32(img.tobytes())
    if size in SYNTH_CRC32 and crc != SYNTH_CRC32[size]:
        raise SystemExit("synthetic %dx%d image has CRC32 %08x, expected %08x (tests/test_abi.py)" % (size, size, crc, SYNTH_CRC32[size]))
    ctx = glf.Context(local_rank)
    if world > 1 or args.force_comm:
        if args.comm == "rccl":   # the library issues its own RCCL collectives: the id travels through torch.distributed once
            # Agree on loadability BEFORE anything collective: every rank probes librccl (glf_rccl_unique_id resolves the
            # library; the id of the others is thrown away), the flags are MIN-reduced, and only then ncclCommInitRank runs
            # everywhere. A failure after that point is fatal: a rank that falls out of a collective init leaves the others
            # inside it.
            ok = 1
            try:
                my_id = glf.rccl_unique_id()
            except Exception as e:                      # librccl not loadable on this rank
                my_id, ok = None, 0
                print("rank %d: %s" % (rank, e), file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=ctx.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                args.comm = "torch"     # every rank takes the same route (and the bench line says so)
                ctx.set_comm_torch(force=args.force_comm)
            else:
                box = [my_id if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                try:
                    ctx.set_comm_rccl(rank, world, box[0], force=args.force_comm)
                except Exception as e:
                    print("rank %d: glf_ctx_set_comm_rccl failed: %s" % (rank, e), file=sys.stderr)
                    os._exit(4)
        else:
            ctx.set_comm_torch(force=args.force_comm)
    d_img = ctx.to_device(img)
    d_out = torch.zeros((size, size), dtype=torch.uint8, device=ctx.device)
    opt = glf.default_options(num_samples=int(N * args.sample_frac), num_eigvals=args.num_eigvals, epsilon=args.epsilon)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run_leg(opt):
        """W untimed + K timed whole-path steps; returns (seconds per step over all ranks, last info,
        mean Nystroem kernel ms, mean stage ms)."""
        info = None
        for _ in range(args.warmup):
            _, _, info = ctx.image_processing(d_img, opt, out=d_out)
        barrier()
        t0 = time.perf_counter()
        kernel_ms = []
        mv = {"launches": 0, "ms": 0.0, "bytes": 0.0, "narrow": 0}
        rp = {"launches": 0, "ms": 0.0, "flops": 0.0}
        cp = {"launches": 0, "ms": 0.0, "flops": 0.0}
        stage_ms = {k: 0.0 for k in ("ms_affinity", "ms_laplacian", "ms_eigen", "ms_nystroem", "ms_filter")}
        for _ in range(args.steps):
            _, _, info = ctx.image_processing(d_img, opt, out=d_out)
            kernel_ms.append(info["nystroem_kernel_ms"])
            mv["launches"] += info["matvecs"]
            mv["narrow"] += info["narrow_sweeps"]
            mv["ms"] += info["matvec_ms"]
            mv["bytes"] += info["matvec_bytes"]
            rp["launches"] += info["nystroem_rowpass_launches"]
            rp["ms"] += info["nystroem_rowpass_ms"]
            rp["flops"] += info["nystroem_rowpass_flops"]
            cp["launches"] += info["nystroem_colpass_launches"]
            cp["ms"] += info["nystroem_colpass_ms"]
            cp["flops"] += info["nystroem_colpass_flops"]
            for k in stage_ms:
                stage_ms[k] += info[k]
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        info["mv"] = mv
        info["rp"] = rp
        info["cp"] = cp
        return elapsed / args.steps, info, float(np.mean(kernel_ms)), {k[3:]: round(v / args.steps, 3) for k, v in stage_ms.items()}

    if world > 1 or args.force_comm:
        ctx.comm_counters(reset=True)
    sec_per_step, info, avg_ms, stage_ms = run_leg(opt)
    ms_per_step = sec_per_step * 1e3
    value = N / sec_per_step * 1e-6
    # what the communicator reports, the collectives of a step, every rank's stage times
    comm_report = None
    if world > 1 or args.force_comm:
        cnt = ctx.comm_counters(reset=True) if args.comm == "rccl" else {}
        ci = ctx.comm_info() if args.comm == "rccl" else {"backend": "torch.distributed callbacks", "size": world, "rccl_ranks": world}
        mine = {"rank": rank, "rows": [info["row0"], info["row1"]], "stage_ms": stage_ms}
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        comm_report = {"backend": ci.get("backend"), "ranks": ci.get("size"), "rccl_ranks": ci.get("rccl_ranks"),
                       "collectives_per_step_rank0": {k: v / (args.steps + args.warmup) for k, v in cnt.items()},
                       "per_rank": per_rank}

    # host_to_host leg (SURVEY 8d's t_e2e: decoded host image bytes -> host output bytes): the image goes up and the result
    # comes down on the context's stream, pinned buffers. Reported beside `value`, which stays the HBM-resident figure.
    host_leg = None
    if world == 1 and not args.no_host_leg:
        h_img = torch.from_numpy(img).pin_memory()
        h_out = torch.empty((size, size), dtype=torch.uint8).pin_memory()
        d_in2 = torch.empty((size, size), dtype=torch.uint8, device=ctx.device)

        def host_step():
            with torch.cuda.stream(ctx.stream):
                d_in2.copy_(h_img, non_blocking=True)
            ctx.image_processing(d_in2, opt, out=d_out)
            with torch.cuda.stream(ctx.stream):
                h_out.copy_(d_out, non_blocking=True)
            ctx.stream.synchronize()
        host_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            host_step()
        barrier()
        h_sec = (time.perf_counter() - t0) / args.steps
        host_leg = {"what": "pinned host image (16.8 MB) -> device, whole path, filtered image -> pinned host buffer; copies on the context's "
                            "stream; PNG codec excluded", "ms_per_step": round(h_sec * 1e3, 3), "value": round(N / h_sec * 1e-6, 4),
                    "unit": "Mpixel/s", "extra_ms_over_hbm_resident": round(h_sec * 1e3 - ms_per_step, 3),
                    "identical_to_hbm_resident_output": bool(torch.equal(h_out, d_out.cpu()))}

    # exact_grid_form leg: the same step with the grid-factored contractions carrying all 256 grey levels through HBM
    exact_leg = rank_leg = None
    if not args.no_exact_leg and info["nystroem_path"] in (3, 4):
        ctx.set_tuning(NYS_PATH="grid", MV_PATH="grid")
        try:
            exact_leg = run_leg(opt)
        finally:
            ctx.set_tuning(NYS_PATH=None, MV_PATH=None)
    # rank_form leg: the same step with the grid-factored contractions in rank form (round 3's first default)
    if not args.no_exact_leg and info["nystroem_path"] == 4:
        ctx.set_tuning(NYS_PATH="rank", MV_PATH="rank")
        try:
            rank_leg = run_leg(opt)
        finally:
            ctx.set_tuning(NYS_PATH=None, MV_PATH=None)
    skip_leg = None
    if not args.no_skip_leg:
        opt_skip = glf.default_options(num_samples=int(N * args.sample_frac), num_eigvals=args.num_eigvals,
                                       epsilon=args.epsilon, skip_exact_zeros=1)
        skip_leg = run_leg(opt_skip)

    # direct_contraction leg: ONE step with the entry-by-entry Nystroem kernel (k_nystroem_f16s: K_B generated in registers,
    # the "true dense contraction" of north_star, SURVEY 8d's W_nys = 2 (N - p) p m) so that it has a driver-timed number
    direct_leg = None
    if not args.no_direct_leg and info["nystroem_path"] in (1, 3, 4):
        ctx.set_tuning(NYS_PATH="direct")
        try:
            ctx.image_processing(d_img, opt, out=d_out)          # warm-up (tables, pool)
            barrier()
            _, _, dinfo = ctx.image_processing(d_img, opt, out=d_out)
            barrier()
            direct_leg = dinfo
        finally:
            ctx.set_tuning(NYS_PATH=None)

    # parity leg input: one more (untimed) run of the headline configuration with its by-products captured
    cap_run = None
    if world == 1 and not args.no_parity and not args.no_cpu_baseline:
        cap_run = ctx.image_processing(d_img, opt, want_float=True, capture=True)
        info["capture"] = cap_run[2]["capture"]

    # nlm leg: the non-local-means affinity (python/affinity_methods/NLM.py) through the whole path on one 1024^2 tile; its K has
    # no factored or banded form (patch distances), every one of the p x N entries is generated twice (degree, extension)
    nlm_leg = None
    if not args.no_nlm_leg and world == 1:
        ts = 1024
        d_nlm = ctx.to_device(glf.synth_image(ts, ts, seed=7))
        o_nlm = glf.default_options(num_samples=int(ts * ts * args.sample_frac), num_eigvals=args.num_eigvals, epsilon=args.epsilon,
                                    kernel=glf.KERNEL_NLM, h_val=3.0)
        ctx.image_processing(d_nlm, o_nlm)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, _, ninfo = ctx.image_processing(d_nlm, o_nlm)
        barrier()
        n_sec = (time.perf_counter() - t0) / args.steps
        entries = float(ninfo["p"]) * ts * ts
        nlm_leg = {"workload": "1024x1024 synthetic, 0.5%% samples (p=%d), m=%d, non-local-means affinity (7x7 patches, h = 3)" % (ninfo["p"], ninfo["m"]),
                   "value": round(ts * ts / n_sec * 1e-6, 3), "unit": "Mpixel/s", "ms_per_image": round(n_sec * 1e3, 3),
                   "stage_ms": {k[3:]: round(ninfo[k], 3) for k in ("ms_affinity", "ms_laplacian", "ms_eigen", "ms_nystroem", "ms_filter")},
                   "kernel_entries_per_pass": entries,
                   "degree_gentries_per_s": round(entries / (ninfo["ms_affinity"] * 1e-3) / 1e9, 1),
                   "nystroem_gentries_per_s": round(entries / (ninfo["ms_nystroem"] * 1e-3) / 1e9, 1),
                   "bound": "vector pipe: 49 weighted patch differences per entry (~100 f32 issue slots per wave and entry; the packed forms issue "
                            "at half rate); the extension's contraction runs on v_mfma_f32_32x32x2_f32",
                   "note": "a next-row feature (SURVEY 8f f1), not the headline path: no MFMA form of the distances yet (DESIGN section 8)"}
        del d_nlm

    batch_leg = None
    if not args.no_batch_leg:
        batch_leg = run_batch_leg(args, ctx, rank, world, barrier)

    if rank == 0:
        p, m = info["p"], info["m"]
        npix_local = (info["row1"] - info["row0"]) * size
        n_samples_local = int(np.count_nonzero((glf.Sampling(size, size, int(N * args.sample_frac)) // size >= info["row0"]) &
                                               (glf.Sampling(size, size, int(N * args.sample_frac)) // size < info["row1"])))
        # algorithmic flops of ONE launch of the Nystroem contraction on this rank:
        # 2 * (non-sample pixels) * p * m   (W_nys = 2 (N - p) p m)
        flops = 2.0 * (npix_local - n_samples_local) * p * m
        achieved = flops / (avg_ms * 1e-3) / 1e12
        ld = 32
        while ld < m:
            ld *= 2
        grid_path = info["nystroem_path"] in (1, 3)
        rank_path = info["nystroem_path"] == 3
        band_path = info["nystroem_path"] == 4
        R = info.get("rank_terms", 0)
        issued = info["nystroem_mfma_flops"] / (avg_ms * 1e-3) / 1e12
        if info["contraction"] == glf.CONTRACT_F16_SPLIT:
            # every f32-equivalent multiply-add is three f16 MFMA products (hi*hi + hi*lo + lo*hi)
            nys_peak = PEAK_F16_MFMA_TFLOPS
            if band_path:
                nys_kernel = ("k_band<%d,2,8,false> (band form: per pixel only the samples within the radius at which 2^15 Er Ec rounds to a "
                              "zero f16 pair -- 211 px at h_loc = 40 -- are evaluated: per (grid row in the band, block of 16 sample columns) the "
                              "32 x 16 entries (Er15 Ec) P(|dv|) are generated from LDS tables, split into f16 hi+lo and contracted with the (hi, lo) "
                              "fragments of Psi, staged by LDS-DMA, on v_mfma_f32_32x32x16_f16; no intermediate reaches HBM)" % (min(ld, 64) // 32))
            elif rank_path:
                nys_kernel = ("k_grid_rowpass_rt<RANK> + k_rank_colpass<%d> (grid-factored Nystroem contraction, rank form: the photometric "
                              "table P(|v - w|) as its rank-%d eigen-expansion F F^T (max error <= 2^-30); S[r][b][k] = sum_a Er f_k(v_ab) Psi "
                              "on v_mfma_f32_32x32x16_f16 with both operands split into f16 hi+lo pairs, then per (image row, segment of <= 32 "
                              "values) T' = F S formed in LDS and Phi = sum_b Ec T' the same way -- T never reaches HBM)" % (R // 16, R))
            elif grid_path:
                nys_kernel = ("k_grid_rowpass_rt<%d> + k_grid_colpass<%d> (grid-factored Nystroem contraction: "
                              "T[r][v][b] = sum_a Er (P Psi) on v_mfma_f32_32x32x16_f16 with both operands split into f16 hi+lo "
                              "pairs, then Phi = sum_b Ec T the same way)" % (ld // 32, ld // 32))
            else:
                nys_kernel = ("k_nystroem_f16s<%d,%d> (direct Nystroem contraction; K_B generated in registers%s, split-f16 MFMA)"
                              % (ld // 32, 2 if ld <= 64 else 1,
                                 " from LDS factor tables" if ld <= 64 and size % 64 == 0 else " with v_exp_f32"))
        else:
            nys_peak = PEAK_F32_MFMA_TFLOPS
            nys_kernel = "k_nystroem<%d,%d> (direct Nystroem contraction, v_mfma_f32_32x32x2_f32)" % (ld // 32, 2 if ld <= 128 else 1)
        nystroem = {
            "kernel": nys_kernel, "path": "band form (NYS_PATH=band)" if band_path else "grid-factored, rank form (NYS_PATH=rank)" if rank_path else "grid-factored" if grid_path else "direct",
            "rank_terms": R,
            "avg_ms_per_step": round(avg_ms, 3), "launches_per_step": info["nystroem_launches"],
            "algorithmic_flops": flops, "algorithmic_tflops": round(achieved, 1),
            "mfma_issued_tflops": round(issued, 1), "mfma_peak_tflops": nys_peak,
            "mfma_issued_frac_of_peak": round(issued / nys_peak, 4),
            "note": "algorithmic = SURVEY 8d W_nys = 2 (N-p) p m flop of the direct contraction; the grid-factored form "
                    "computes the same sums with 2 H 256 nc nr m + 2 N nc m flop, so algorithmic_tflops may exceed the MFMA "
                    "peak; mfma_issued counts the f16 MFMA flops actually executed (x3 for the split products)",
        }
        # ---- the L_A sweeps of the eigensolver, and the dominant kernel of the step ---------------------------
        mvs, rps = info["mv"], info["rp"]
        mv_avg_ms = mvs["ms"] / max(1, mvs["launches"])
        prof = {}
        try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this configuration
            key = "%dx%d_m%d_%s_gpus%d" % (size, size, m, "f16s" if info["contraction"] == glf.CONTRACT_F16_SPLIT else "f32", n_gpus)
            prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key, {})
        except (OSError, ValueError):
            pass
        sweeps = {"path": ("band form (L_A never stored: k_band<.,1,8,true> on the samples within the radius of each sample; Y = alpha (D X - K_A X) in its epilogue)" if info["matvec_path"] == 4 else
                           "grid-factored, rank form (L_A never stored; S = 0.7 GB per sweep instead of T = 5.6 GB)" if info["matvec_path"] == 3 else
                           "grid-factored (L_A never stored)" if info["matvec_path"] == 1 else "stored L_A streamed (k_block_matvec_f16s)"),
                  "launches_per_step": mvs["launches"] / args.steps, "avg_ms": round(mv_avg_ms, 4),
                  "ms_per_step": round(mvs["ms"] / args.steps, 3),
                  "narrow_per_step": mvs["narrow"] / args.steps,
                  "narrow": "sweeps applied to the still-iterating columns only, packed into a block of 32 (block PCG)"}
        # (image row, value) pairs that occur in the image: the T rows the row pass has to deliver (the others are computed
        # by the 32-row MFMA tiles but never stored or read)
        present = float(np.mean([np.unique(img[r]).size for r in range(info["row0"], info["row1"])])) / 256.0
        cps = info["cp"]
        if band_path and cps["launches"] > 0:
            # dominant kernel of the band form: the Nystroem launch of k_band (the L_A sweeps run the same kernel on the samples)
            cp_avg_ms = cps["ms"] / cps["launches"]
            cp_flops = cps["flops"] / cps["launches"]
            cp_tflops = cp_flops / (cp_avg_ms * 1e-3) / 1e12
            cp_peak = PEAK_F16_MFMA_TFLOPS / 3.0
            ev_flops = 2.0 * info["nystroem_evaluated"] * min(ld, 64)
            roofline = {
                "kernel": "k_band<%d,2,8,false> (Phi[px][n] = sum over the samples within the radius of (Er15 Ec P)[px][s] Psi[s][n]: entries "
                          "generated on the vector pipe from LDS tables and split into f16 hi+lo, v_mfma_f32_32x32x16_f16 with the Psi "
                          "fragments staged by LDS-DMA, f32 accumulate; 8 waves = 8 image rows x 64 columns per workgroup)" % (min(ld, 64) // 32),
                "bound": "mfma", "achieved": round(cp_tflops, 1), "peak": round(cp_peak, 1), "unit": "TFLOP/s",
                "frac": round(cp_tflops / cp_peak, 4), "traffic": prof.get("band_bytes_per_launch"),
                "traffic_source": prof.get("band_source", "none: no committed --pmc pass for this configuration"),
                "avg_launch_ms": round(cp_avg_ms, 4), "launches_per_step": cps["launches"] / args.steps,
                "flops_per_launch": cp_flops, "ms_per_step": round(cps["ms"] / args.steps, 3),
                "evaluated_tflops": round(ev_flops / (cp_avg_ms * 1e-3) / 1e12, 1),
                "evaluated_frac_of_peak": round(ev_flops / (cp_avg_ms * 1e-3) / 1e12 / cp_peak, 4),
                "peak_basis": "f16 dense MFMA peak 2500 TFLOP/s / 3 products per split-precision multiply-add (nominal)",
                "note": "achieved = algorithmic flops of the launch -- 2 m x the (pixel, sample) pairs whose distance is inside the radius "
                        "(the circle), one product per multiply-add -- / mean HIP-event duration of the launch. The entries evaluated are "
                        "~1.8x that (whole blocks of 16 sample columns x 64-pixel tiles cover the circle: evaluated_*). traffic = FETCH_SIZE x 2 "
                        "+ WRITE_SIZE of the launch from the committed rocprofv3 --pmc passes named in traffic_source: the Psi fragments "
                        "(24 MB, L2-resident) + the Phi write"}
        elif rank_path and cps["launches"] > 0:
            # dominant kernel of the rank form: the fused T' + column pass of the Nystroem stage (the L_A sweeps run the same kernel
            # over the nr grid rows)
            cp_avg_ms = cps["ms"] / cps["launches"]
            cp_flops = cps["flops"] / cps["launches"]
            cp_tflops = cp_flops / (cp_avg_ms * 1e-3) / 1e12
            cp_peak = PEAK_F16_MFMA_TFLOPS / 3.0
            rp_avg_ms = rps["ms"] / max(1, rps["launches"])
            roofline = {
                "kernel": "k_rank_colpass<%d,false> (per image row and segment of <= 32 grey values: T'[v][b][n] = sum_k f_k(v) S[r][b][k][n] "
                          "formed block by block of 16 sample columns in LDS, then Phi[px][n] = sum_b Ec(c_px - C_b) T'[v_px][b][n], both on "
                          "v_mfma_f32_32x32x16_f16 with operands split into f16 hi+lo, f32 accumulate; persistent workgroups)" % (R // 16),
                "bound": "mfma", "achieved": round(cp_tflops, 1), "peak": round(cp_peak, 1), "unit": "TFLOP/s",
                "frac": round(cp_tflops / cp_peak, 4), "traffic": prof.get("rank_colpass_bytes_per_launch"),
                "traffic_source": prof.get("rank_colpass_source", "none: no committed --pmc pass for this configuration"),
                "algorithmic_bytes": 4.0 * (N * (ld if ld < 64 else 64) * (ld // min(ld, 64)) + info["p"] // max(1, int(np.sqrt(info["p"]))) * 0) ,
                "avg_launch_ms": round(cp_avg_ms, 4), "launches_per_step": cps["launches"] / args.steps,
                "flops_per_launch": cp_flops, "ms_per_step": round(cps["ms"] / args.steps, 3),
                "row_pass": {"kernel": "k_grid_rowpass_rt<1,.,4,2,8,.,RANK> (S[r][b][k] = sum_a Er(r - R_a) f_k(v_ab) Psi[(a,b)])",
                             "avg_launch_ms": round(rp_avg_ms, 4), "launches_per_step": rps["launches"] / args.steps,
                             "tflops": round(rps["flops"] / max(1e-9, rps["ms"] * 1e-3) / 1e12, 1),
                             "frac_of_peak": round(rps["flops"] / max(1e-9, rps["ms"] * 1e-3) / 1e12 / cp_peak, 4)},
                "peak_basis": "f16 dense MFMA peak 2500 TFLOP/s / 3 products per split-precision multiply-add (nominal)",
                "note": "achieved = algorithmic flops of the launch -- 2 ld ncs (sum over rows of (values present) R + pixels): T' for the "
                        "(row, value) pairs that occur, then the Ec contraction per pixel; one product per multiply-add -- / mean HIP-event "
                        "duration of the launch. The MFMAs executed are ~2.5x that (32-value and 32-pixel tiles are part-filled: ~23 of 32 "
                        "slots, ~19 of 32 pixels per chunk at cfg4). traffic = FETCH_SIZE x 2 + WRITE_SIZE of the launch from the committed "
                        "rocprofv3 --pmc passes named in traffic_source: the S stream (read once from HBM, re-read ~9x from L2) + the Phi write"}
            roofline.pop("algorithmic_bytes")
        elif info["matvec_path"] == 1 and rps["launches"] > 0:
            # dominant kernel: the row pass of the Nystroem contraction (k_grid_rowpass_rt; by name the largest share of the step)
            rp_avg_ms = rps["ms"] / rps["launches"]
            rp_flops = rps["flops"] / rps["launches"]
            rp_tflops = rp_flops / (rp_avg_ms * 1e-3) / 1e12
            rp_peak = PEAK_F16_MFMA_TFLOPS / 3.0
            roofline = {
                "kernel": "k_grid_rowpass_rt<%d> (T[r][v][b] = sum_a Er(r - R_a) (P(|v - v_ab|) Psi[(a,b)]): per (sample column b, "
                          "value v) a GEMM [rows x grid rows] x [grid rows x m] on v_mfma_f32_32x32x16_f16, both operands split into "
                          "f16 hi+lo, f32 accumulate; the launches of the Nystroem stage -- the L_A sweeps of the eigensolver compute "
                          "the same sums over the grid rows with k_grid_rowpass)" % (min(ld, 64) // 32),
                "bound": "mfma", "achieved": round(rp_tflops, 1), "peak": round(rp_peak, 1), "unit": "TFLOP/s",
                "frac": round(rp_tflops / rp_peak, 4), "traffic": prof.get("grid_rowpass_bytes_per_launch"),
                "traffic_source": prof.get("grid_rowpass_source", "none: no committed --pmc pass for this configuration"),
                "useful_work_frac": round(present, 4), "frac_useful": round(rp_tflops / rp_peak * present, 4),
                "peak_basis": "f16 dense MFMA peak 2500 TFLOP/s / 3 products per split-precision multiply-add (nominal; "
                              "tools/mfma16_probe.hip: v_mfma_f32_32x32x16_f16 sustains 1750 TFLOP/s from registers and 1670 with "
                              "B fragments from LDS on pseudo-random operands, 2450 on all-zero operands -- power-limited)",
                "avg_launch_ms": round(rp_avg_ms, 4), "launches_per_step": rps["launches"] / args.steps,
                "flops_per_launch": rp_flops, "ms_per_step": round(rps["ms"] / args.steps, 3),
                "note": "achieved = algorithmic 2 rows 256 nc nr m flop of the launch (one product per multiply-add) / mean "
                        "HIP-event duration of the launch; traffic = its T write (FETCH_SIZE x 2 + WRITE_SIZE) from the committed "
                        "rocprofv3 --pmc passes named in traffic_source (PMC counters cannot be read from inside this process); "
                        "useful_work_frac = (image row, value) pairs that occur / all 256 per row: frac_useful counts only those"}
        else:
            mv_bytes = mvs["bytes"] / max(1, mvs["launches"])          # algorithmic: 4 p (rows of this rank) per launch
            mv_gbs = mv_bytes / (mv_avg_ms * 1e-3) / 1e9 if mv_avg_ms > 0 else 0.0
            roofline = {
                "kernel": "k_block_matvec_f16s<%d> (block mat-vec of the PCG / residual: L_A streamed once per launch, "
                          "split-f16 MFMA contraction with the %d-column block)" % (ld // 32, ld),
                "bound": "hbm", "achieved": round(mv_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(mv_gbs / PEAK_HBM_GBS, 4), "traffic": prof.get("matvec_bytes_per_launch"),
                "traffic_source": prof.get("matvec_source", "none: no committed --pmc pass for this configuration"),
                "avg_launch_ms": round(mv_avg_ms, 4), "launches_per_step": mvs["launches"] / args.steps,
                "bytes_per_launch": mv_bytes, "ms_per_step": round(mvs["ms"] / args.steps, 3),
                "note": "achieved = algorithmic 4 p rows bytes (the L_A block of this rank) / mean HIP-event duration of the "
                        "sweep kernel; the largest single-kernel share of the step"}
        line = {
            "metric": "filtered Mpixels/sec @ 4K img, 0.5% samples; PSNR vs PETSc ref",
            "value": round(value, 4), "unit": "Mpixel/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("f32 (f32 storage and accumulation; MFMA operands split into f16 hi+lo pairs = 22 significant bits)"
                      if info["contraction"] == glf.CONTRACT_F16_SPLIT else "f32"),
            "data": "synthetic",
            "config": {"workload": "%dx%d synthetic noisy image, %.1f%% samples (p=%d), m=%d eigenpairs, eps=%g, "
                                   "no exact-zero skipping (every kernel entry enters the sums: %s)" % (size, size, args.sample_frac * 100, p, m, args.epsilon,
                                   "band form: the samples beyond the radius contribute exact zeros in this arithmetic" if band_path else
                                   "grid-factored forms, rank-%d photometric expansion" % R if rank_path else
                                   "grid-factored forms" if info["nystroem_path"] == 1 else "entry-by-entry kernels"),
                       "N": N, "p": p, "m": m, "epsilon": args.epsilon, "outer_its": info["outer_its"],
                       "inner_its_total": info["inner_its_total"], "residual": round(info["residual"], 5),
                       "contraction": "f16 split (hi+lo), f32 accumulate" if info["contraction"] == glf.CONTRACT_F16_SPLIT else "f32 MFMA",
                       "image_crc32": "%08x" % crc,
                       "psnr_reference": "PETSc/SLEPc are not installable here (SURVEY 8c): PSNR is against the fp64 restatement of "
                                         "hpc/*.c, see cpu_baseline.parity_headline (this run, sampled rows) and .parity_cfg2 (whole image)",
                       "sharding": "pixel rows / %d ranks; eigen-solve %s; collectives: %s" %
                                   (n_gpus, ("rows / %d ranks (all-reduce of inner products and Gram blocks, all-gather of the operand per L_A "
                                             "application)" % n_gpus) if info.get("eigen_sharded", 0) or n_gpus == 1 else
                                    "replicated on every rank (band form: a sweep costs less than the all-gather of its operand; one all-reduce "
                                    "of D_A and the value-weighted sums per image)", "none" if n_gpus == 1 and not args.force_comm else
                                    ("RCCL issued by the library" if args.comm == "rccl" else "torch.distributed callbacks"))},
            "stage_ms_rank0": stage_ms,
            "filter_fused": bool(info.get("filter_fused", 0)),
            "roofline": roofline,
            "eigen_sweeps": sweeps,
            "nystroem": nystroem,
        }
        if comm_report is not None:
            line["comm"] = comm_report
            line["rccl_ranks"] = comm_report.get("rccl_ranks")
        if host_leg is not None:
            line["host_to_host"] = host_leg
        if exact_leg is not None:
            e_sec, e_info, e_avg_ms, e_stage = exact_leg
            line["exact_grid_form"] = {
                "what": "same workload with GLF_NYS_PATH = GLF_MV_PATH = grid: the grid-factored contractions carrying all 256 grey levels "
                        "(T = 78 GB per image through HBM)",
                "value": round(N / e_sec * 1e-6, 4), "unit": "Mpixel/s", "ms_per_step": round(e_sec * 1e3, 3), "stage_ms_rank0": e_stage,
                "outer_its": e_info["outer_its"], "inner_its_total": e_info["inner_its_total"]}
        if rank_leg is not None:
            r_sec, r_info, r_avg_ms, r_stage = rank_leg
            line["rank_form"] = {
                "what": "same workload with GLF_NYS_PATH = GLF_MV_PATH = rank: the grid-factored contractions with the photometric table as "
                        "its rank-%d expansion (S = 10 GB per image through HBM instead of T = 78 GB; T' formed in LDS)" % r_info.get("rank_terms", 0),
                "value": round(N / r_sec * 1e-6, 4), "unit": "Mpixel/s", "ms_per_step": round(r_sec * 1e3, 3), "stage_ms_rank0": r_stage,
                "outer_its": r_info["outer_its"], "inner_its_total": r_info["inner_its_total"]}
        if skip_leg is not None:
            s_sec, s_info, s_avg_ms, s_stage = skip_leg
            dense_evals = float(p) * npix_local
            line["exact_zero_skipping"] = {
                "what": "same workload with glf_options.skip_exact_zeros = 1: tiles whose kernel entries are exactly 0 in the "
                        "arithmetic in use are not evaluated; output bit-identical to the headline run (tests/test_gpu_parity.py)",
                "value": round(N / s_sec * 1e-6, 4), "unit": "Mpixel/s", "ms_per_step": round(s_sec * 1e3, 3),
                "stage_ms_rank0": s_stage, "nystroem_kernel_ms": round(s_avg_ms, 3),
                "fraction_evaluated": {"nystroem": round(s_info["nystroem_evaluated"] / dense_evals, 4),
                                       "degree": round(s_info["degree_evaluated"] / dense_evals, 4)},
                "outer_its": s_info["outer_its"],
            }
        if batch_leg is not None:
            line["throughput_mode"] = batch_leg
        if nlm_leg is not None:
            line["nlm"] = nlm_leg
        if direct_leg is not None:
            d_ms = direct_leg["nystroem_kernel_ms"]
            d_tf = flops / (d_ms * 1e-3) / 1e12
            line["direct_contraction"] = {
                "kernel": "k_nystroem_f16s<%d,%d> (Phi = K_B^T Psi entry by entry: K_B generated in registers from LDS factor tables, "
                          "split-f16 MFMA, f32 accumulate; GLF_NYS_PATH=direct)" % (ld // 32, 2 if ld <= 64 else 1),
                "steps": 1, "kernel_ms": round(d_ms, 3), "step_ms": round(direct_leg["ms_total"], 3),
                "roofline": {"bound": "mfma", "achieved": round(d_tf, 1), "peak": round(PEAK_F16_MFMA_TFLOPS / 3.0, 1), "unit": "TFLOP/s",
                             "frac": round(d_tf / (PEAK_F16_MFMA_TFLOPS / 3.0), 4), "traffic": None,
                             "flops_per_launch": flops,
                             "note": "achieved = SURVEY 8d W_nys = 2 (N - p) p m flop / HIP-event duration of the kernel; peak = f16 "
                                     "dense MFMA peak / 3 split products"}}
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(img, info, args)
        if n_gpus == 1 and not args.no_parity and "cpu_baseline" in line:
            if cap_run is not None:
                line["cpu_baseline"]["parity_headline"] = parity_headline(ctx, img, cap_run, args)
            line["cpu_baseline"]["parity_cfg2"] = cpu_parity_cfg2(ctx)
        line.pop("capture", None)
        print(json.dumps(line))
    ctx.close()
    if world > 1 or args.force_comm:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
