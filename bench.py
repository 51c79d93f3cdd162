#!/usr/bin/env python3
"""bench.py -- images/sec of N-step Natural Inference on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: BASELINE config 2, i.e. 15-step Natural
Inference with ``weights/step_15_weight_173.npz`` on a batch of 512 CIFAR10-shaped samples: 15 NCSN++
forwards in the HIP engine (bf16 MFMA) + 15 fused ``ni_step`` launches (fp64 history, the reference's
arithmetic).  Inputs (noise, weights, coefficient rows) are resident in HBM before the timed region.
Consecutive steps alternate between two HIP streams (two engine handles, two history buffers), the order
``CIFAR10NaturalInference.natural_inference_tx`` runs its batches in: the images are bit-identical to the
one-after-the-other order, whose time for the same K steps is reported as ``single_stream`` (``--streams 1`` makes it the headline).

    python bench.py --gpus N --steps K --warmup W

Multi-GPU: generation batches are independent, so rank r simply runs its own batches (weak scaling, no
collective on the data path); timing is barrier + synchronize on both sides and the max over ranks.  One
process per GPU: under ``torch.distributed.run`` (WORLD_SIZE set) this process IS a rank; a bare
``python bench.py --gpus N`` starts the N ranks itself (``launch_ranks``: a child ``torch.distributed.run``,
started before this process has touched the GPU -- the reference uses every visible GPU from one command,
deps/score_sde_pytorch/models/utils.py:93) and relays rank 0's JSON line.

The default line carries all three north-star workloads: the CIFAR10 metric at top level plus ``sd3`` (BASELINE
config 4) and ``sd3_fp8`` (config 5) objects with their own value / ms_per_step / rooflines / cpu_baseline
(``--no-sd3`` skips them; ``--workload sd3 [--fp8]`` runs one of them alone as the top-level line).

The JSON line also carries
  roofline          the dominant kernel (k_conv_gn2, MFMA-bound): algorithmic flops per launch / mean launch
                    duration, measured with HIP events on the engine's stream over an instrumented one-stream replica of
                    the timed region (one kernel on the GPU at a time)
  roofline_gemm     the same for the remaining k_gemm_* launches; roofline_whole_denoiser: all flops / all device time
  roofline_ni_step  the named recurrence kernel (HBM-bound): algorithmic bytes per launch / mean duration
  cpu_baseline      the CPU oracle (eager PyTorch restatement of the reference path) timed on this host (BASELINE.md section 3:
                    config 1 and a B=64 15-step point, denoiser / combine split)
  accuracy          image-level difference of the bf16 engine's 15-step samples from the fp32 oracle's on identical noise
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                # HBM3E spec, same guide
GFLOP_PER_IMAGE_FORWARD = 21.69307136  # 2*MAC of conv/linear/attention matmuls (oracle.flops_per_image)


def ni_step_bytes_per_element(C, s_x=4, s_h=8):
    """reads s_x*(2 + nnzB) + s_h*(nnzA-1), writes s_h + s_x, per element per step
    (zero-skipped rows; B has one non-zero per row in weights/step_*)."""
    import numpy as np
    out = []
    for k in range(C.shape[0]):
        nnz_a = int(np.count_nonzero(C[k, :k + 1]))
        out.append(s_x * (2 + 1) + s_h * (nnz_a - 1) + s_h + s_x)
    return out


def profiled_traffic(match, exclude=None):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary (profiles/rNN/*_hbm_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE passes over this very command, gfx950 x2 read correction applied by
    tools/summarize_profile.py).  bench.py cannot run rocprofv3 itself; returns None when no summary exists."""
    files = sorted(ROOT.glob("profiles/r*/*_hbm_traffic.json"))
    if not files:
        return None, None
    tab = json.loads(files[-1].read_text())
    excl = (exclude,) if isinstance(exclude, str) else tuple(exclude or ())
    rows = [v for k, v in tab.items() if match in k and not any(e in k for e in excl)]
    n = sum(v["launches"] for v in rows)
    if not n:
        return None, None
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows) / n, str(files[-1].relative_to(ROOT))


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks as a CHILD ``torch.distributed.run`` (this parent has not
    touched the GPU and never does), pass the child's output through and exit with its code.  Rank 0's JSON line is
    re-printed last on stdout, everything else goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines, last_json = p.stdout.splitlines(), None
    for ln in lines:
        if ln.startswith("{") and '"metric"' in ln:
            last_json = ln
        else:
            print(ln, file=sys.stderr)
    if last_json is not None:
        print(last_json, flush=True)
    if p.returncode != 0 or last_json is None:
        raise SystemExit(p.returncode or 1)


def timed_region(one_step, steps, warmup, world, sync, dist, dev):
    """W untimed steps, then exactly K steps between (synchronize, barrier, synchronize) brackets; returns the max over ranks (s)."""
    import torch

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()
    out = None
    for i in range(warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        out = one_step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    return dt, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--no-single-stream", dest="single_stream_extra", action="store_false", help="skip the extra one-stream timing of the CIFAR10 workload")
    ap.add_argument("--streams", type=int, default=2, help="CIFAR10 workload: HIP streams the consecutive batches (steps) alternate between, as in natural_inference_tx "
                                                          "(default 2; 1 = one batch after the other)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--weights", default=str(ROOT / "weights" / "step_15_weight_173.npz"))
    ap.add_argument("--workload", choices=["cifar10", "sd3", "selftest"], default="cifar10",
                    help="cifar10 = the BASELINE.json metric (default; its line also carries the sd3 / sd3_fp8 objects); sd3 = config 4 alone "
                         "(SD3 1024x1024 28-step NI, MMDiT bf16); selftest = the launcher / rendezvous / timing skeleton without kernels (CPU, gloo)")
    ap.add_argument("--fp8", action="store_true", help="with --workload sd3: BASELINE config 5 (sharp-variant weights, fp8 e4m3 GEMM operands)")
    ap.add_argument("--no-sd3", action="store_true", help="default workload: leave out the sd3 / sd3_fp8 objects")
    ap.add_argument("--sd3-steps", type=int, default=2, help="timed 4-image batches of each SD3 configuration inside the default line")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process-group backend (gloo: the selftest workload on CPU)")
    ap.add_argument("--same-device", action="store_true",
                    help="functional test of the N > 1 path on a ONE-GPU box: every rank uses cuda:0 (gloo backend only: RCCL refuses two ranks on one device)")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="selftest: this rank exits non-zero (launcher error-path test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the event-instrumented replica (for rocprof runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)              # before anything here imports torch or touches a GPU

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    if args.workload == "selftest":
        return bench_selftest(args, world, rank)
    if args.same_device:
        if args.backend != "gloo":
            raise SystemExit("--same-device needs --backend gloo")
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.workload == "sd3":
        line = bench_sd3(args, world, rank, dev, fp8=args.fp8, steps=args.steps, warmup=args.warmup)
    else:
        line = bench_cifar(args, world, rank, dev)
        if not args.no_sd3:
            # configs 4 / 5 in the same line: every rank runs them (they shard like the CIFAR10 batches), rank 0 reports
            import gc
            gc.collect(); torch.cuda.empty_cache()
            from naturaldiffusion_amd.mmdit import SD3_MEDIUM
            from naturaldiffusion_amd.synth import synthetic_mmdit_flat
            flat = synthetic_mmdit_flat(grid=64, seed=0, **SD3_MEDIUM)
            for key, fp8 in (("sd3", False), ("sd3_fp8", True)):
                sub = bench_sd3(args, world, rank, dev, fp8=fp8, steps=args.sd3_steps, warmup=1, flat=flat)
                for drop in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data"):
                    sub.pop(drop, None)
                line[key] = sub
                gc.collect(); torch.cuda.empty_cache()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_selftest(args, world, rank):
    """The multi-rank skeleton of this file with no kernels in it: rendezvous on 127.0.0.1, barrier-bracketed timed region, max over
    ranks, rank 0 prints the line.  tests/test_bench_launcher.py drives ``python bench.py --gpus 2 --workload selftest --backend gloo``."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend if args.backend == "gloo" else "gloo")
    if rank == args.selftest_fail_rank:
        raise SystemExit(3)
    a = torch.ones(64, 64)
    dt, out = timed_region(lambda i: (a @ a).sum() + rank, args.steps, args.warmup, world, lambda: None, dist, torch.device("cpu"))
    tot = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(tot)
    if rank == 0:
        print(json.dumps({"metric": "selftest steps/sec (launcher + rendezvous + timing skeleton, no kernels)", "value": round(world * args.steps / dt, 2),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "selftest", "backend": "gloo", "rank_sum": float(tot[0])}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_cifar(args, world, rank, dev):
    import numpy as np
    import torch
    import torch.distributed as dist
    from naturaldiffusion_amd import _lib
    from naturaldiffusion_amd.coeff import load_coeff_npz
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.sampler import CifarNI
    from naturaldiffusion_amd.synth import synthetic_flat_params

    _lib.require_gpu()
    Bz = args.batch
    C, Bm, node = load_coeff_npz(args.weights)
    n_step = node.shape[0] - 1
    E = Bz * 3 * 32 * 32
    flat = synthetic_flat_params(0)
    engine = NCSNppEngine(flat, max_batch=Bz, device=dev)
    ni = CifarNI(C, Bm, node, E, device=dev)
    gen = torch.Generator(device=dev).manual_seed(888 + rank)
    noises = [torch.randn(Bz, 3, 32, 32, generator=gen, device=dev) for _ in range(2)]

    def one_step(i):
        return ni.run(engine, noises[i & 1])

    # The job is a sequence of independent batches (50,000 images = ranks x batches of 512; reference loop CIFAR10NaturalInference.py:287-309).  As in
    # natural_inference_tx, consecutive batches (= steps) go to `--streams` HIP streams, each with its own engine handle and history buffer: a step is still
    # one 15-step pass over one batch of 512, exactly K of them are timed between the synchronize / barrier brackets, and the under-occupied launches of one
    # batch run under the other's convolutions.  The single-stream time of the same K steps is measured as well and reported next to it.
    n_str = max(1, min(args.streams, args.steps))
    lanes = [(engine, ni, None)] + [(NCSNppEngine(flat, max_batch=Bz, device=dev), CifarNI(C, Bm, node, E, device=dev), None) for _ in range(n_str - 1)]
    if n_str > 1:
        lanes = [(e_, n_, torch.cuda.Stream(device=dev)) for e_, n_, _ in lanes]

    def one_step_streams(i):
        e_, n_, st = lanes[i % n_str]
        with torch.cuda.stream(st):
            return n_.run(e_, noises[i & 1])

    # set-up, not a step: one forward per lane on its own stream (first use of the handle's workspace; the kernels' code objects load on first launch)
    for e_, n_, st in lanes:
        with torch.cuda.stream(st):
            e_(noises[0], torch.full((Bz,), 500.0, device=dev))
    torch.cuda.synchronize()
    dt1 = None
    if n_str == 1 or args.single_stream_extra:
        dt1, out = timed_region(one_step, args.steps, args.warmup, world, torch.cuda.synchronize, dist, dev)
        assert torch.isfinite(out).all()
    if n_str > 1:
        dt, out = timed_region(one_step_streams, args.steps, args.warmup, world, torch.cuda.synchronize, dist, dev)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
    else:
        dt = dt1
    imgs = world * Bz * args.steps
    value = imgs / dt

    line = {
        "metric": "images/sec at 15-step Natural Inference (CIFAR10 32x32, NCSN++)",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "CIFAR10 Natural Inference 15-step (step_15_weight_173.npz), batch=512 per GPU, "
                               "NCSN++ (cifar10_ddpmpp_continuous, 61.8M params, synthetic weights) bf16 MFMA / fp32 acc, "
                               "ni_step fp64 history",
                   "coeff_file": os.path.basename(args.weights), "nfe": n_step, "batch_per_gpu": Bz,
                   "sharding": f"batch-sharded x{world}, no collective",
                   "streams": n_str},
    }
    if n_str > 1 and dt1 is not None:
        line["single_stream"] = {"value": round(imgs / dt1, 2), "ms_per_step": round(dt1 / args.steps * 1e3, 3),
                                 "note": "the same K steps one batch after the other on one HIP stream (the reference's order).  The headline alternates consecutive batches "
                                         "between two streams (two engine handles; bit-identical images): the under-occupied launches of one batch run under the other's "
                                         "convolutions.  The roofline objects below are measured in THIS order -- kernel durations are only meaningful with one kernel "
                                         "on the GPU at a time"}

    if rank == 0 and not args.no_roofline:
        # ---- instrumented replica of the timed region: HIP events around every engine launch group and
        # ---- every ni_step launch (same stream)
        engine.profile(True)
        ev = []
        orig_step = ni.step

        def timed_step(k, *a, **kw):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            r = orig_step(k, *a, **kw)
            a1.record()
            ev.append((k, a0, a1))
            return r
        ni.step = timed_step
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(i)
        torch.cuda.synchronize()
        dt_inst = time.perf_counter() - t1
        prof = engine.profile_read()
        engine.profile(False)
        ni.step = orig_step
        gemm_ms, gemm_n = prof["gemm"]
        other_ms, other_n = prof["other"]
        cg_ms, cg_n = prof["conv_gn"]
        c8_ms, c8_n = prof["conv_gn8"]
        fwd = args.steps * n_step
        # algorithmic flops per launch family: 2*M*N*K of every matmul-shaped launch the plan issues at this batch
        # (natinf_ncsnpp_describe_gemms); their sum is the 21.69 GFLOP / image / forward of SURVEY section 8d
        rows = engine.describe_gemms(Bz)
        fl = lambda r: 2.0 * r[0] * r[1] * (r[2] + r[3]) * r[5]
        is8 = lambda r: r[0] in (Bz * 64, Bz * 16)                                # the 8x8 and 4x4 levels: 64 / 16 pixels per image
        cg_rows = [r for r in rows if r[6].startswith("conv_gn") and not is8(r)]
        c8_rows = [r for r in rows if r[6].startswith("conv_gn") and is8(r)]
        cg_flops = sum(fl(r) for r in cg_rows) * fwd
        c8_flops = sum(fl(r) for r in c8_rows) * fwd
        gemm_flops = GFLOP_PER_IMAGE_FORWARD * 1e9 * Bz * fwd - cg_flops - c8_flops          # the rest: the k_gemm_* launches, the fused attention, the head
        all_ms = cg_ms + c8_ms + gemm_ms + other_ms
        ach = cg_flops / (cg_ms * 1e-3) / 1e12
        tr_cg, tr_src = profiled_traffic("k_conv_gn", exclude=("k_conv_gn2<8", "k_conv_gn2<4"))
        line["roofline"] = {
            "kernel": "k_conv_gn2<32 | 16, ...> (3x3 convolution with GroupNorm-apply + SiLU fused into its operand path, weights streamed through registers; the 32x32 and "
                      f"16x16 levels: the dominant kernel, {100 * cg_ms / all_ms:.0f} % of the engine's device time; its 8x8 / 4x4 instantiations: roofline_conv_gn8)", "bound": "mfma",
            "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": tr_cg, "traffic_source": tr_src, "traffic_note": "HBM bytes per launch from the committed rocprofv3 PMC summary named in traffic_source (the same launches), not measured by this run",
            "launches": int(cg_n), "mean_launch_ms": round(cg_ms / cg_n, 5),
            "flops_per_launch": cg_flops / cg_n, "device_ms_total": round(cg_ms, 3),
            "ms_per_step_instrumented": round(dt_inst / args.steps * 1e3, 3),
        }
        if c8_n:
            ach8 = c8_flops / (c8_ms * 1e-3) / 1e12
            line["roofline_conv_gn8"] = {
                "kernel": "k_conv_gn2<8 | 4, true, ..., 4> (the same fused kernel on the 8x8 and 4x4 levels: 64-pixel x 256-channel tiles = one 8x8 image, two blocks per CU / "
                          f"four 4x4 images, two K groups of four waves per block, 128 tiles at B = 512; {100 * c8_ms / all_ms:.0f} % of the engine's device time)", "bound": "mfma", "achieved": round(ach8, 2),
                "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach8 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "launches": int(c8_n),
                "mean_launch_ms": round(c8_ms / c8_n, 5), "flops_per_launch": c8_flops / c8_n, "device_ms_total": round(c8_ms, 3)}
        ach_g = gemm_flops / (gemm_ms * 1e-3) / 1e12
        tr_gemm, tr_src_g = profiled_traffic("k_gemm")
        line["roofline_gemm"] = {
            "kernel": "k_gemm_* (LDS-DMA implicit-GEMM tile variants: resampling-block convolutions, NIN, linear) + the 16x16 attention block's k_qkv256 / k_attn256<true> (HBM-bound) + k_head_conv",
            "bound": "mfma", "achieved": round(ach_g, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach_g / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": tr_gemm, "traffic_source": tr_src_g, "launches": int(gemm_n),
            "mean_launch_ms": round(gemm_ms / gemm_n, 5), "flops_per_launch": gemm_flops / gemm_n, "device_ms_total": round(gemm_ms, 3),
            "other_kernels_device_ms": round(other_ms, 3), "other_launches": int(other_n),
        }
        line["roofline_whole_denoiser"] = {
            "bound": "mfma", "achieved": round((cg_flops + c8_flops + gemm_flops) / (all_ms * 1e-3) / 1e12, 2),
            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round((cg_flops + c8_flops + gemm_flops) / (all_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "note": "all matmul flops of the forward / device time of ALL its kernels (normalisation, statistics, softmax included)"}
        bpe = ni_step_bytes_per_element(C)
        tot_ms = sum(a0.elapsed_time(a1) for _, a0, a1 in ev)
        tot_bytes = sum(bpe[k] * E for k, _, _ in ev)
        ach_gbs = tot_bytes / (tot_ms * 1e-3) / 1e9
        tr_ni, tr_src = profiled_traffic("k_step_f64hist")
        line["roofline_ni_step"] = {
            "kernel": "k_step_f64hist", "bound": "hbm", "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach_gbs / HBM_PEAK_GBS, 4), "traffic": tr_ni, "traffic_source": tr_src,
            "launches": len(ev), "mean_launch_ms": round(tot_ms / len(ev), 5),
            "bytes_per_launch": tot_bytes / len(ev), "bytes_per_element_by_step": bpe,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (eager-PyTorch restatement of the reference path, fp32 NCSN++ +
        # ---- fp64 recurrence) on this host's cores; two bounded points, each split into denoiser and combine time
        from oracle import ni_oracle as O, ncsnpp_oracle as N
        from naturaldiffusion_amd.synth import synthetic_state_dict
        try:
            import psutil
            physical = psutil.cpu_count(logical=False) or os.cpu_count()
        except Exception:
            physical = os.cpu_count()
        # BASELINE.md asks for threads = physical cores; at these problem sizes (8 .. 64 images per forward) eager PyTorch on CPU
        # scales to ~32 threads and gets SLOWER beyond (OpenMP barriers per op: measured 2.1 images/s at 32 threads, 1.3 at 128
        # on this host class in round 1), so the baseline is given its best configuration: min(32, physical)
        threads = max(1, min(32, physical, torch.get_num_threads()))
        torch.set_num_threads(threads)
        P = synthetic_state_dict(0)
        model = N.model_fn_from_params(P)
        spent = {"t": 0.0}

        def timed_model(x, labels):
            t_ = time.perf_counter()
            r = model(x, labels)
            spent["t"] += time.perf_counter() - t_
            return r
        model(torch.zeros(1, 3, 32, 32), torch.zeros(1))                   # page in / warm the thread pool
        points = []
        final64 = None
        for wname, nb in (("step_5_weight_00.npz", 8), (os.path.basename(args.weights), 64)):
            Cc, Bc, nodec = load_coeff_npz(ROOT / "weights" / wname) if (ROOT / "weights" / wname).exists() else (C, Bm, node)
            z = torch.randn(nb, 3, 32, 32, generator=torch.Generator().manual_seed(888))
            spent["t"] = 0.0
            tc = time.perf_counter()
            xs = O.cifar_ni_trajectory(timed_model, z, Cc, Bc, nodec)
            dc = time.perf_counter() - tc
            points.append({"coeff_file": wname, "images": nb, "nfe": int(nodec.shape[0] - 1), "seconds": round(dc, 2),
                           "images_per_s": round(nb / dc, 4), "denoiser_s": round(spent["t"], 2), "combine_s": round(dc - spent["t"], 3)})
            if nb == 64:
                final64, z64 = xs[-1], z
        main_pt = points[-1]
        line["cpu_baseline"] = {"value": main_pt["images_per_s"], "unit": "images/s", "cores": threads, "kind": "port",
                                "sample": f"{main_pt['images']} images x {main_pt['nfe']} steps, same coefficient file and synthetic weights "
                                          f"(fp32 NCSN++ oracle + fp64 recurrence), {main_pt['seconds']} s "
                                          f"({main_pt['denoiser_s']} s denoiser + {main_pt['combine_s']} s combine); host: {physical} physical / "
                                          f"{os.cpu_count()} logical CPUs, {threads} threads (see bench.py: more threads are slower at this size)",
                                "physical_cores": physical, "points": points}
        # ---- accuracy of the bf16 engine at the image level (stand-in for the FID delta, which is blocked on assets): the same
        # ---- 64 noise tensors through the HIP path, against the fp32 / fp64 oracle trajectory just computed
        if final64 is not None and os.path.basename(args.weights) == "step_15_weight_173.npz":
            eng64 = NCSNppEngine(synthetic_flat_params(0), max_batch=64, device=dev)
            got = CifarNI(C, Bm, node, 64 * 3 * 32 * 32, device=dev).run(eng64, z64.to(dev)).cpu()
            d = (got - final64).abs()
            pg, pr = O.to_pixel(got).to(torch.int16), O.to_pixel(final64).to(torch.int16)
            pd = (pg - pr).abs()
            rms = lambda t: float((t.double() ** 2).mean().sqrt())
            line["accuracy"] = {"what": "final x of 15-step NI (step_15_weight_173), the 64 images of the cpu_baseline sample, HIP bf16 engine vs the fp32 "
                                        "oracle on identical noise.  The synthetic network is not a denoiser (|x| reaches 1e3), so the numbers are relative; "
                                        "tests/test_gpu_accuracy.py has the image-level figures with a well-conditioned denoiser (256 images: mean |dx| "
                                        "0.0075, 0.67 uint8 steps per pixel) and shows the error equals that of an fp32-accumulate model with bf16 operands",
                                "rel_rms": round(rms(got - final64) / rms(final64), 5), "rel_max": round(float(d.max() / final64.abs().max()), 5),
                                "x_abs_max": round(float(final64.abs().max()), 1), "uint8_pixels_differing": round(float((pd > 0).float().mean()), 4),
                                "uint8_mean_abs_diff": round(float(pd.float().mean()), 4),
                                "fid": "blocked: checkpoint_8.pth / Inception weights / cifar10_mu_sigma.npz absent"}

    del engine, ni
    return line


def sd3_reduced_depth_accuracy(dev, weights_csv="sd3_step_28_weight_sharp.csv", n=2, seed=3):
    """What fp8 operands cost over a whole 28-step SD3-form Natural Inference run, at a depth the fp32 CPU oracle finishes in seconds:
    a 4-block / 256-wide MMDiT (4 heads x 64, 16x16 image tokens + 29 text tokens, 16-channel 32x32 latents), CFG 7, the shipped
    coefficient file -- final latents of the HIP engine with bf16 operands and with NATINF_MMDIT_FP8 against oracle/mmdit_oracle.py
    (fp32; PARITY UNPINNED) + the oracle's restatement of the loop (src/SD3NaturalInference.py:198-223) on identical noise.
    Checker code: only bench.py's accuracy leg and tests/ call this."""
    import numpy as np
    import torch
    from oracle import mmdit_oracle as MO, ni_oracle as O
    from naturaldiffusion_amd.coeff import load_sd3_csv
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    from naturaldiffusion_amd.sampler import SD3NI
    cfg = dict(layers=4, heads=4, joint_dim=128, pooled_dim=64)
    grid, tc, nstep = 16, 29, 28
    P = MO.make_params(seed=11, pos_max=32, pos_base=16, **cfg)
    P = dict(P); P["proj_out.weight"] = P["proj_out.weight"] * 0.2          # O(1) velocities: a well-conditioned 28-step fp16 chain
    g = torch.Generator().manual_seed(seed)
    pe, ne = torch.randn(n, tc, 128, generator=g), torch.randn(n, tc, 128, generator=g)
    ppe, npe = torch.randn(n, 64, generator=g), torch.randn(n, 64, generator=g)
    noises = torch.randn(n, 16, 2 * grid, 2 * grid, generator=g).half()
    W = load_sd3_csv(ROOT / "weights" / weights_csv)
    ts, sig = O.sd3_sigma_schedule(nstep)

    def vel(x, t, cond):
        tt = torch.as_tensor(t, dtype=torch.float32).expand(n)
        return MO.forward(P, x.float(), tt, pe if cond else ne, ppe if cond else npe).half()
    ref = O.sd3_ni(vel, noises, W, sig, ts).float()
    flat = flatten_state_dict(P, grid, **cfg)
    text, pooled = torch.cat([pe, ne]).to(dev), torch.cat([ppe, npe]).to(dev)
    out = {}
    for name, fp8 in (("bf16", False), ("fp8", True)):
        eng = MMDiTEngine(flat, max_batch=2 * n, grid=grid, ctx_tokens=tc, device=dev, fp8=fp8, **cfg)
        ni = SD3NI(W, sig.to(dev), noises.numel(), device=dev, cfg=7.0)
        zflat = noises.to(dev).reshape(-1)
        x = ni.first_input(zflat)
        for k in range(nstep):
            xx = x.view(n, 16, 2 * grid, 2 * grid)
            v = eng.forward(torch.cat([xx, xx]), ts[k].to(dev).expand(2 * n), text, pooled)
            mean, x = ni.step(k, x, v[:n].reshape(-1), v[n:].reshape(-1), zflat, want_next=k + 1 < nstep)
        got = mean.view(n, 16, 2 * grid, 2 * grid).float().cpu()
        d = got - ref
        out[name] = {"rel_rms": round(float((d ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()), 5), "rel_max": round(float(d.abs().max() / ref.abs().max()), 5),
                     "mean_abs": round(float(d.abs().mean()), 5), "finite": bool(torch.isfinite(got).all())}
        del eng
    out["what"] = (f"final latents of 28-step SD3-form NI ({weights_csv}, CFG 7, {n} images, 32x32x16 latents) through a 4-block / 256-wide MMDiT: HIP engine "
                   "(bf16 operands | fp8 e4m3 image-stream GEMM operands) vs the fp32 oracle + oracle loop on identical noise; latents rms "
                   f"{float((ref ** 2).mean().sqrt()):.3f}, max {float(ref.abs().max()):.3f}")
    return out


def bench_sd3(args, world, rank, dev, fp8=False, steps=2, warmup=1, flat=None):
    """BASELINE config 4: SD3NaturalInference 28-step (weights/sd3_step_28_weight.csv), 1024x1024 (latents
    [4,16,128,128] fp16), CFG 7 -> per step ONE batched MMDiT forward of 8 sequences (4,096 image + 333 text tokens) +
    one fused natinf_step_f16chain launch.  A "step" of this bench = one 4-image batch through all 28 steps.
    SD3-medium-shaped synthetic weights and synthetic text embeddings (the checkpoint / text encoders are downloads)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from naturaldiffusion_amd import _lib
    from naturaldiffusion_amd.coeff import load_sd3_csv
    from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
    from naturaldiffusion_amd.sampler import SD3NI
    from naturaldiffusion_amd.synth import synthetic_mmdit_flat
    _lib.require_gpu()
    n, tc, nstep = 4, 333, 28
    wname = "sd3_step_28_weight_sharp.csv" if fp8 else "sd3_step_28_weight.csv"
    W = load_sd3_csv(ROOT / "weights" / wname)
    u = np.linspace(1.0, 3 * 0.001 / (1 + 2 * 0.001), nstep)          # FlowMatchEulerDiscreteScheduler, shift 3
    sig = np.append(3 * u / (1 + 2 * u), 0.0).astype(np.float32)
    sigmas, timesteps = torch.from_numpy(sig).to(dev), torch.from_numpy(sig[:-1] * 1000).to(dev)
    if flat is None:
        flat = synthetic_mmdit_flat(grid=64, seed=0, **SD3_MEDIUM)
    eng = MMDiTEngine(flat, max_batch=2 * n, grid=64, ctx_tokens=tc, device=dev, fp8=fp8, **SD3_MEDIUM)
    g = torch.Generator(device=dev).manual_seed(10 + rank)
    noises = torch.randn(n, 16, 128, 128, device=dev, dtype=torch.float16, generator=g)
    text = torch.randn(2 * n, tc, 4096, device=dev, generator=g)
    pooled = torch.randn(2 * n, 2048, device=dev, generator=g)
    ni = SD3NI(W, sigmas, noises.numel(), device=dev, cfg=7.0)
    zflat = noises.reshape(-1)

    def one_step():
        x = ni.first_input(zflat)
        for k in range(nstep):
            xx = x.view(n, 16, 128, 128)
            v = eng.forward(torch.cat([xx, xx]), timesteps[k].expand(2 * n), text, pooled)
            mean, x = ni.step(k, x, v[:n].reshape(-1), v[n:].reshape(-1), zflat, want_next=k + 1 < nstep)
        return mean

    dt, out = timed_region(lambda i: one_step(), steps, warmup, world, torch.cuda.synchronize, dist, dev)
    assert torch.isfinite(out.float()).all()
    D, L, tx = 1536, 24, 4096
    T = tx + tc
    flops_fwd_seq = L * (2.0 * T * 3 * D * D + 4.0 * T * T * D + 2.0 * T * D * D + 2.0 * T * 8 * D * D) - 2.0 * tc * 9 * D * D + 2.0 * tc * 4096 * D
    tf = flops_fwd_seq * 2 * n * nstep * steps / dt / 1e12
    line = {"metric": "images/sec at 28-step Natural Inference (SD3 1024x1024, MMDiT)", "value": round(world * n * steps / dt, 4),
            "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8+bf16" if fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"SD3 Natural Inference 28-step ({wname}), 4 images x CFG per GPU = 8 sequences of 4096+333 "
                                   "tokens per forward, SD3-medium-shaped MMDiT (2.03 B params, synthetic weights) "
                                   + ("fp8 e4m3 operands for the image-stream q|k, v, fc1 GEMMs, bf16 elsewhere, fp32 acc, " if fp8 else "bf16 MFMA / fp32 acc, ")
                                   + "ni_step fp16 chain", "nfe": 2 * nstep, "images_per_gpu": n, "sharding": f"batch-sharded x{world}, no collective"},
            "roofline": {"kernel": "whole forward (k_gemm_dma / k_gemm_fp8 + k_flash_attn64), 2*MAC flops / wall time; peak = dense bf16", "bound": "mfma", "achieved": round(tf, 1),
                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}}
    if rank == 0 and not args.no_roofline:
        # ---- per-kernel rooflines, measured live with HIP events on the kernels' own ABI entry points at the engine's shapes
        # ---- (the engine has no per-launch hook; these are the same kernels, same shapes, same process)
        from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
        ev = lambda: torch.cuda.Event(enable_timing=True)

        def timed(fn, iters=5):
            fn(); torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            for _ in range(iters):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / iters * 1e-3
        Bs, H, Tp = 2 * n, 24, (T + 127) // 128 * 128
        q = torch.randn(Bs, Tp, 2 * D, device=dev).bfloat16()
        vT = torch.randn(Bs, D, Tp, device=dev).bfloat16()
        o = torch.empty(Bs, Tp, D, device=dev, dtype=torch.bfloat16)
        t_fa_iso = timed(lambda: check(lib.natinf_attention_hd64_bf16(ptr(q), ptr(q) + 2 * D, 2 * D, Tp * 2 * D, ptr(vT), ptr(o), D, Tp * D, Bs, H, Tp, T, 0.125,
                                                                       stream_ptr()), "attention"))
        # the same kernel where it actually runs: one more 28-step batch with an event pair around every k_flash_attn64 launch of the
        # engine (natinf_attention_profile).  Round 2 quoted the isolated back-to-back loop above (1.44 ms) next to rocprof's in-engine
        # average (1.15 ms): both were right -- a loop of nothing but this kernel draws more power than the GEMM / attention mix of a
        # forward and gets a lower clock on this power-capped part (rocprofv3 shows the same 1.2 vs 1.4-1.5 ms split, profiles/r03)
        import ctypes
        check(lib.natinf_attention_profile(1), "attention_profile")
        one_step(); torch.cuda.synchronize()
        ms_tot, n_l = ctypes.c_double(), ctypes.c_int64()
        check(lib.natinf_attention_profile_read(ctypes.byref(ms_tot), ctypes.byref(n_l)), "attention_profile_read")
        check(lib.natinf_attention_profile(0), "attention_profile")
        t_fa = ms_tot.value / max(1, n_l.value) * 1e-3
        fa_flops = 4.0 * T * T * 64 * H * Bs
        line["roofline"] = {"kernel": "k_flash_attn64 (joint attention, 8 sequences x 24 heads x 4,429 keys, head_dim 64), timed in the engine: HIP events "
                                      "around each of its launches in one 28-step batch", "bound": "mfma",
                            "achieved": round(fa_flops / t_fa / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(fa_flops / t_fa / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                            "mean_launch_ms": round(t_fa * 1e3, 4), "launches": int(n_l.value), "flops_per_launch": fa_flops,
                            "isolated_loop_ms": round(t_fa_iso * 1e3, 4), "launches_per_image_batch": L * nstep,
                            "share_of_forward_flops": round(L * fa_flops / (flops_fwd_seq * Bs), 3)}
        M = Bs * tx
        shapes = [("q|k", 2 * D, D), ("v^T / out", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)]
        tot_t, tot_f, tot_ideal, per = 0.0, 0.0, 0.0, {}
        for name, N_, K_ in shapes:
            if fp8 and name != "v^T / out":
                a8 = torch.randint(0, 255, (M, K_), device=dev, dtype=torch.uint8); b8 = torch.randint(0, 255, (N_, K_), device=dev, dtype=torch.uint8)
                a8 &= 0x77; b8 &= 0x77                                                 # finite e4m3 patterns
                sa, sb = torch.ones(M, device=dev), torch.ones(N_, device=dev)
                c = torch.empty(M, N_, device=dev, dtype=torch.bfloat16)
                t_ = timed(lambda: check(lib.natinf_debug_gemm_fp8(M, N_, K_, ptr(a8), ptr(sa), None, ptr(b8), ptr(sb), None, ptr(c), None, 0, 1, stream_ptr()), "gemm_fp8"))
                kind = "fp8"
            else:
                a = torch.randn(M, K_, device=dev).bfloat16(); b = torch.randn(N_, K_, device=dev).bfloat16()
                c = torch.empty(M, N_, device=dev, dtype=torch.bfloat16)
                t_ = timed(lambda: check(lib.natinf_debug_gemm(0, M, N_, K_, 0, 1, 0, 1, ptr(a), None, ptr(b), None, ptr(c), 0, 1.0, 1, stream_ptr()), "gemm"))
                kind = "bf16"
            f_ = 2.0 * M * N_ * K_
            pk_ = 5000.0 if kind == "fp8" else MFMA_BF16_PEAK_TFLOPS
            per[name] = {"M": M, "N": N_, "K": K_, "operands": kind, "ms": round(t_ * 1e3, 4), "TFLOP/s": round(f_ / t_ / 1e12, 1),
                         "peak": pk_, "frac": round(f_ / t_ / 1e12 / pk_, 4)}
            tot_t += t_; tot_f += f_; tot_ideal += f_ / (pk_ * 1e12)
        # one fraction over mixed operand types: the time the dense peaks of each shape's own operand type would need / the time taken
        line["roofline_gemm"] = {"kernel": ("k_gemm_fp8 (image-stream q|k, fc1, fc2: e4m3 operands, v_mfma_f32_16x16x128_f8f6f4; peak 5,000) + k_gemm_dma (v^T / out: bf16; peak 2,500)" if fp8
                                            else "k_gemm_dma<2,4,8,4> 256x256 LDS-DMA tiles") + ", image-stream projections of one block",
                                 "bound": "mfma", "achieved": round(tot_f / tot_t / 1e12, 1), "peak": round(tot_f / tot_ideal / 1e12, 1), "unit": "TFLOP/s",
                                 "frac": round(tot_ideal / tot_t, 4), "traffic": None, "shapes": per,
                                 "note": "peak = flop-weighted dense peak of the shapes' operand types; frac = sum(flops_i / peak_i) / sum(time_i)"}
        line["roofline_whole_forward"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "note": "2*MAC flops of all matmuls / wall time of the bench (dense bf16 peak)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the MMDiT oracle (fp32, eager PyTorch) on ONE sequence through TWO blocks at full width, extrapolated to the
        # ---- 24 blocks x 2 sequences (CFG) x 28 steps of an image -- a full forward on the CPU would take minutes
        from oracle import mmdit_oracle as MO
        threads = max(1, min(32, torch.get_num_threads()))
        torch.set_num_threads(threads)
        cfgo = dict(layers=2, heads=24, joint_dim=4096, pooled_dim=2048)
        Po = MO.make_params(seed=1, pos_max=64, pos_base=64, **cfgo)
        gg = torch.Generator().manual_seed(3)
        xo, to_ = torch.randn(1, 16, 128, 128, generator=gg), torch.tensor([500.0])
        eo, po = torch.randn(1, tc, 4096, generator=gg), torch.randn(1, 2048, generator=gg)
        tcpu = time.perf_counter()
        MO.forward(Po, xo, to_, eo, po)
        d2 = time.perf_counter() - tcpu
        per_image_s = d2 / 2 * L * 2 * nstep
        line["cpu_baseline"] = {"value": round(1.0 / per_image_s, 6), "unit": "images/s", "cores": threads, "kind": "port",
                                "sample": f"oracle/mmdit_oracle.py (fp32): one sequence (4096+333 tokens) through 2 of the 24 blocks in {d2:.1f} s, "
                                          f"extrapolated x{L // 2} blocks x 2 sequences (CFG) x {nstep} steps = {per_image_s:.0f} s per image; host has {os.cpu_count()} logical CPUs"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del eng
        line["accuracy"] = sd3_reduced_depth_accuracy(dev, wname)
        eng = None
    del eng, ni
    return line


if __name__ == "__main__":
    main()
