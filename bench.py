#!/usr/bin/env python3
"""bench.py -- images/sec of N-step Natural Inference on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload cifar10|sd3|fid50k|validate]

The JSON line carries NUMBERS and short identifiers only (the whole line stays under 3 KB so that it survives any log tail);
what every field means is written HERE.

Top level (default workload = BASELINE config 2).  One "step" = one pass of the hot path over one batch: 15-step Natural
Inference with ``weights/step_15_weight_173.npz`` on 512 CIFAR10-shaped samples = 15 NCSN++ forwards in the HIP engine (bf16
MFMA operands, fp32 accumulate) + 15 fused ``ni_step`` launches (fp64 history, the reference's arithmetic).  Inputs (noise,
weights, coefficient rows) are resident in HBM before the timed region.  ``value`` is MULTI-STREAM THROUGHPUT: consecutive steps
rotate over ``--streams`` HIP streams (three since round 6, two before: that many engine handles sharing one copy of the packed weights, one history buffer each) -- the order
``CIFAR10NaturalInference.natural_inference_tx`` / ``generate_sharded`` run their batches in; the images are bit-identical to the
one-after-the-other order.  ``ms_per_step`` = time / K, an inverse throughput.
  single_stream     {value, ms_per_step}: the same K steps one batch after the other on ONE stream (the reference's order; the
                    like-for-like number for rounds 1-2, whose headline was this).  ``--streams 1`` makes it the headline.
  pipeline          {value}: images/s of the same K batches (K x 512 images) through the production pipeline (``generate_sharded`` on the same lanes): Philox noise by
                    global index drawn inside the timed span, the 15 steps, ``to_pixel``, and the ONE copy of the uint8 images to the host -- everything the
                    reference's per-batch loop contains (:290, :308-309) that the contract's "inputs resident in HBM" region leaves out.
  sd3, sd3_fp8      BASELINE configs 4 / 5 (``--workload sd3 [--fp8]`` alone): 28-step SD3-form NI at 1024x1024, 4 images x CFG per
                    GPU = ONE batched MMDiT forward of 8 sequences (4,096 image + 333 text tokens; inside it the text stream's launches run on a HIP
                    stream of the engine's own, joined with the image stream at every joint attention: same bytes; the image tokens' residual stream in IEEE half since
                    round 6 -- the reference's pipeline is fp16 --, natinf_set_mmdit_stream16) per step + one fused
                    ``natinf_step_f16chain`` launch; a step = one 4-image batch through all 28 steps; SD3-medium-shaped synthetic
                    weights (2.03 B parameters).  fp8 = e4m3 operands (v_mfma_f32_16x16x128_f8f6f4) for the image-stream q|k, v, fc1,
                    fc2 GEMMs, bf16 elsewhere.  Fields: value (images/s), ms_per_step, frac (all 2*MAC flops of the forward / wall time
                    / 2,500 TFLOP/s), roofline {k_flash_attn64_v2 timed IN the engine: HIP events around each launch of one 28-step batch: ms,
                    TFLOP/s, frac of the bf16 peak; iso_ms = the same kernel in a back-to-back loop, which this power-capped part clocks
                    lower}, gemm {per image-stream projection role (qk, vT, out, fc1, fc2; 4,096 tokens x 8 sequences), timed IN the engine by natinf_gemm_profile --
                    the kernel with the epilogue the engine gives it: [TFLOP/s, frac of ITS operand type's dense peak (bf16 2,500 / fp8 5,000), mean launch us];
                    frac = sum(flops_i / peak_i) / sum(time_i); iso_fc1 = fc1 with that same epilogue in an isolated back-to-back loop, TFLOP/s}, cpu (the MMDiT oracle on one sequence through 2
                    of 24 blocks, extrapolated: images/s), acc (28-step NI through a 4-block / 256-wide MMDiT: relative RMS of the final
                    latents vs the fp32 oracle, both precisions, in the `sd3` object only -- PARITY UNPINNED, the oracle restates the published architecture).
  fid50k            BASELINE config 3 (``--workload fid50k`` alone): the 50,000-image FID job of reference
                    src/CIFAR10NaturalInference.py:281-312 for BOTH coefficient-matrix equivalents -- DPM-Solver++(2S)
                    (results/dpmsolverpp/dpmsolverpp2s_018.npz) and DDIM on the continuous VP grid (coeffgen.ddim_vp_continuous over
                    linspace(1, 1e-3, 19)), 18 NFE each.  This rank generates ITS share (global indices rank, rank + W, ...; batches of
                    512 + one ragged batch) with ``generate_sharded`` (two lanes, Philox noise by global index, uint8 images kept on
                    the device), scores it with the Inception-V3 pool3 engine (500 images per call: the reference's 50 give the same features, a quarter slower), and
                    ``calc_fid_sharded`` sums (n, sum, outer-product sum) over ranks with ONE all-reduce and evaluates the Frechet
                    distance (the trace term through two symmetric eigen-decompositions, in float64 on the GPU: fid_stats.frechet_distance; pytorch_fid's scipy sqrtm form on the host gives the same number in 5-14 s;
                    the reference statistics' side of it is taken once per job, and rank 0 evaluates each distance on a host thread while the next matrix's images are generated).  With ONE GPU the default share is rank 0 of 8
                    (``share_of``: 6,250 images; ``--fid-share-of 1`` runs all 50,000).  value = images generated AND scored per second
                    over all ranks, both matrices; s = wall seconds {gen, inception, allreduce, frechet = the host threads' own time, frechet_wait = what of it the step
                    still waits for at its end} summed over the two matrices;
                    gen_rate = images/s of generation alone; inc_rate = Inception images/s.  The checkpoint, the Inception weights and
                    cifar10_mu_sigma.npz are downloads: without them synthetic weights / (0, I) reference statistics stand in and
                    fid = "blocked"; d_matrices = Frechet distance between the two matrices' image statistics (same network, same
                    noise: a sanity figure that needs no asset).
  validate          SURVEY 8f N2 + N4 (``--workload validate`` alone): ``ValidateNaturalInference.natural_inference("ddim", 24)`` --
                    DiT-XL/2 engine (residual stream in IEEE half since round 6: natinf_set_dit_stream16), 8 class-conditional latents, CFG 4 (the conditional and unconditional calls of a step as ONE forward of 16), 24 steps, one fused
                    ``natinf_step_f32prod`` launch per step -- then the AutoencoderKL decoder engine (8 x 256x256 images) and the PNG
                    row.  value = images/s; dit_ms = mean DiT forward (16 samples); vae_ms = decode of the 8 latents; synthetic weights.
  roofline          the dominant kernel class, ``k_conv_gn2`` / ``k_conv_gn3`` at 32x32 / 16x16 (3x3 convolution with GroupNorm-apply + SiLU fused into its
                    operand path; k_conv_gn3 -- one wave per SIMD, 128 x 128 wave tiles -- takes the long-K launches; MFMA-bound): achieved = algorithmic flops per launch (2*M*N*K of the launches, from the engine's own
                    launch table) / mean launch duration, measured with HIP events on the engine's stream over an instrumented ONE-stream
                    replica of the timed region (one kernel on the GPU at a time); traffic = HBM bytes per launch from the newest
                    committed rocprofv3 PMC summary (profiles/rNN/*_hbm_traffic.json: bench.py cannot run rocprofv3 on itself);
                    share = its part of the engine's device time.
  roofline_conv_gn8 the same kernel's 8x8 / 4x4 instantiations; roofline_gemm: every other matmul-shaped launch (k_gemm_*, the 16x16
                    attention block k_attn_blk256(_v2), k_head_conv; the flops are the REFERENCE's -- four projections per attention block: the v2 kernel multiplies two of folded matrices);
                    roofline_whole_denoiser: all flops / all device time.
  roofline_ni_step  ``k_step_f64hist`` (HBM-bound): algorithmic bytes per launch / mean launch duration.
  cpu_baseline      the CPU oracle (eager-PyTorch restatement of the reference path: fp32 NCSN++ + fp64 recurrence) on this host:
                    64 images x 15 steps; ``config1`` = BASELINE config 1 (8 images x 5 steps) images/s; threads = min(32, physical):
                    more are slower at this size.
  accuracy          final x of the 64 cpu_baseline images, HIP bf16 engine vs fp32 oracle on identical noise: relative RMS, mean
                    |delta| in uint8 steps.  (The synthetic network is not a denoiser: tests/test_gpu_accuracy.py has the image-level
                    figures with a well-conditioned one.)  fid: blocked on checkpoint_8.pth / Inception weights / cifar10_mu_sigma.npz.

Multi-GPU: generation batches are independent, so rank r runs its own batches (weak scaling, no collective on the data path;
fid50k: one all-reduce of statistics at the end); timing is barrier + synchronize on both sides and the max over ranks.  One process
per GPU: under ``torch.distributed.run`` (WORLD_SIZE set) this process IS a rank; a bare ``python bench.py --gpus N`` starts the N
ranks itself (``launch_ranks``, before this process touches a GPU) and relays rank 0's line.  With N > 1 only rank 0 runs the
instrumented replica / CPU legs, and the SD3 / validate sub-objects are left to ``--workload`` runs (the line then carries fid50k).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the first HIP call (naturaldiffusion_amd/_lib.py says why): streams that share a hardware queue do not overlap

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500         # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000
HBM_PEAK_GBS = 8000                  # HBM3E spec, same guide
GFLOP_PER_IMAGE_FORWARD = 21.69307136  # 2*MAC of conv/linear/attention matmuls (oracle.flops_per_image)
INCEPTION_GFLOP_PER_IMAGE = 11.42      # pool3 path at 299x299


def r4(x):
    """4 significant digits: the line is numbers, keep them short"""
    if x is None:
        return None
    f = float(f"{float(x):.4g}")
    return int(f) if f == int(f) and abs(f) >= 100 else f          # (1236 instead of 1236.0: ~100 bytes of the line)


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
    excl = (exclude,) if isinstance(exclude, str) else tuple(exclude or ())
    for f in sorted(ROOT.glob("profiles/r*/*_hbm_traffic.json"), reverse=True):      # newest round first; within it the first summary that holds the kernel (SD3 / CIFAR10 / Inception runs are separate files)
        tab = json.loads(f.read_text())
        rows = [v for k, v in tab.items() if match in k and not any(e in k for e in excl)]
        n = sum(v["launches"] for v in rows)
        if n:
            return r4(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows) / n)
    return None


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


def timed_region(one_step, steps, warmup, world, sync, dist, dev, keep=2):
    """W untimed steps, then exactly K steps between (synchronize, barrier, synchronize) brackets; returns the max over ranks (s)
    and the outputs of the last ``keep`` steps (one per lane of a multi-stream run)."""
    import torch

    grouped = dist.is_available() and dist.is_initialized()      # world > 1, or one rank under --force-pg (the RCCL rehearsal of a one-GPU box)

    def barrier():
        sync()
        if grouped:
            dist.barrier()
        sync()
    outs = []
    for i in range(warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        outs = (outs + [one_step(i)])[-keep:]
    barrier()
    dt = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    return dt, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 8 for cifar10, 2 for sd3, 1 for fid50k, 3 for validate)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 1; 0 for fid50k)")
    ap.add_argument("--no-single-stream", dest="single_stream_extra", action="store_false", help="skip the extra one-stream timing of the CIFAR10 workload")
    ap.add_argument("--streams", type=int, default=3, help="CIFAR10 / fid50k: HIP streams (lanes) the consecutive batches rotate over (default 3 since round 6: +1 % over 2 in same-box A/Bs, profiles/r06/streams_ab.txt; 1 = one batch after the other)")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--weights", default=str(ROOT / "weights" / "step_15_weight_173.npz"))
    ap.add_argument("--workload", choices=["cifar10", "sd3", "fid50k", "validate", "selftest"], default="cifar10",
                    help="cifar10 = the BASELINE.json metric (default; its line also carries sd3 / sd3_fp8 / fid50k / validate objects); the others alone "
                         "as the top-level line; selftest = the launcher / rendezvous / timing skeleton without kernels (CPU, gloo)")
    ap.add_argument("--fp8", action="store_true", help="with --workload sd3: BASELINE config 5 (sharp-variant weights, fp8 e4m3 GEMM operands)")
    ap.add_argument("--no-sd3", action="store_true", help="default workload: leave out the sd3 / sd3_fp8 objects")
    ap.add_argument("--no-fid50k", action="store_true", help="default workload: leave out the fid50k object")
    ap.add_argument("--no-validate", action="store_true", help="default workload: leave out the validate object")
    ap.add_argument("--sd3-steps", type=int, default=2, help="timed 4-image batches of each SD3 configuration inside the default line")
    ap.add_argument("--fid-samples", type=int, default=50000, help="fid50k: images of the whole job (reference: 50,000)")
    ap.add_argument("--fid-share-of", type=int, default=0, help="fid50k on ONE GPU: run rank 0's share of a job sharded this many ways (default 8; 1 = the whole job)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process-group backend (gloo: the selftest workload on CPU)")
    ap.add_argument("--force-pg", action="store_true",
                    help="initialise the process group even with ONE rank and run every collective of the N > 1 path through it (barriers, the max-over-ranks "
                         "all-reduce of a device tensor, fid50k's 33.6 MB fp64 all-reduce): with --backend nccl the RCCL rehearsal a one-GPU box can do")
    ap.add_argument("--same-device", action="store_true",
                    help="functional test of the N > 1 path on a ONE-GPU box: every rank uses cuda:0 (gloo backend only: RCCL refuses two ranks on one device)")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="selftest: this rank exits non-zero (launcher error-path test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the event-instrumented replica (for rocprof runs)")
    args = ap.parse_args()
    dflt = {"cifar10": (8, 1), "sd3": (2, 1), "fid50k": (1, 0), "validate": (3, 1), "selftest": (8, 1)}[args.workload]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)              # before anything here imports torch or touches a GPU

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
    grouped = world > 1 or args.force_pg
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:            # --force-pg without a launcher: a one-rank rendezvous of our own
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import gc

    def release():
        gc.collect(); torch.cuda.empty_cache()

    if args.workload == "sd3":
        line = bench_sd3(args, world, rank, dev, fp8=args.fp8, steps=args.steps, warmup=args.warmup)
    elif args.workload == "fid50k":
        line = bench_fid50k(args, world, rank, dev, steps=args.steps, warmup=args.warmup)
    elif args.workload == "validate":
        line = bench_validate(args, world, rank, dev, steps=args.steps, warmup=args.warmup)
    else:
        line = bench_cifar(args, world, rank, dev)
        tail = {k: line.pop(k) for k in list(line) if k.startswith(("config", "roofline", "cpu_baseline", "accuracy"))}
        strip = ("metric", "unit", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "config", "warmup", "steps")
        drop = ("bound", "peak", "unit", "traffic", "flops_per_launch", "launches", "flop_share", "sample", "kind", "where")       # constants of the sub-objects: in the docstring

        def slim(o):
            return {k: slim(v) for k, v in o.items() if k not in drop} if isinstance(o, dict) else o
        release()
        if not args.no_sd3 and world == 1:
            # configs 4 / 5 in the same line (one GPU: with N > 1 they are `--workload sd3 [--fp8]` runs -- every rank would repeat them)
            from naturaldiffusion_amd.mmdit import SD3_MEDIUM
            from naturaldiffusion_amd.synth import synthetic_mmdit_flat
            flat = synthetic_mmdit_flat(grid=64, seed=0, **SD3_MEDIUM)
            for key, fp8 in (("sd3", False), ("sd3_fp8", True)):
                sub = bench_sd3(args, world, rank, dev, fp8=fp8, steps=args.sd3_steps, warmup=1, flat=flat)
                line[key] = slim({k: v for k, v in sub.items() if k not in strip and k != "dtype" and not (fp8 and k == "acc")})      # (acc carries both precisions: once, in "sd3")
                release()
            del flat
        if not args.no_fid50k:
            sub = bench_fid50k(args, world, rank, dev, steps=1, warmup=0)
            line["fid50k"] = slim({k: v for k, v in sub.items() if k not in strip and k not in ("dtype", "cpu_baseline", "batches", "last_batch", "frechet_vs_synthetic_ref", "ms_per_step")})
            release()
        if not args.no_validate and world == 1:
            sub = bench_validate(args, world, rank, dev, steps=3, warmup=1)
            line["validate"] = slim({k: v for k, v in sub.items() if k not in strip and k not in ("dtype", "cpu_baseline", "dit_forwards")})
            release()
        line.update(tail)
    if grouped:
        line.setdefault("config", {})["process_group"] = f"{dist.get_backend()} x{world}"
    if rank == 0:
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if grouped:
        dist.destroy_process_group()


def bench_selftest(args, world, rank):
    """The multi-rank skeleton of this file with no kernels in it: rendezvous on 127.0.0.1, barrier-bracketed timed region, max over
    ranks, rank 0 prints the line.  tests/test_bench_launcher.py drives ``python bench.py --gpus 2 --workload selftest --backend gloo``."""
    import torch
    import torch.distributed as dist
    grouped = world > 1 or args.force_pg
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:            # --force-pg without a launcher: a one-rank rendezvous of our own (as main() does for the GPU workloads)
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(args.backend if args.backend == "gloo" else "gloo")
    if rank == args.selftest_fail_rank:
        raise SystemExit(3)
    a = torch.ones(64, 64)
    dt, _ = timed_region(lambda i: (a @ a).sum() + rank, args.steps, args.warmup, world, lambda: None, dist, torch.device("cpu"))
    tot = torch.tensor([float(rank + 1)])
    if grouped:
        dist.all_reduce(tot)
    if rank == 0:
        print(json.dumps({"metric": "selftest steps/sec (launcher + rendezvous + timing skeleton, no kernels)", "value": round(world * args.steps / dt, 2),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "selftest", "backend": "gloo", "rank_sum": float(tot[0]),
                                     "process_group": (f"{dist.get_backend()} x{dist.get_world_size()}" if grouped else None)}}), flush=True)
    if grouped:
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

    # The job is a sequence of independent batches (reference loop CIFAR10NaturalInference.py:287-309).  As in natural_inference_tx /
    # generate_sharded, consecutive batches (= steps) go to `--streams` HIP streams, each with its own engine handle (cloned: shared
    # packed weights) and history buffer: a step is still one 15-step pass over one batch of 512, exactly K of them are timed.
    n_str = max(1, min(args.streams, args.steps))
    lanes = [(engine, ni, None)] + [(engine.clone(), CifarNI(C, Bm, node, E, device=dev), None) for _ in range(n_str - 1)]
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
        dt1, outs = timed_region(one_step, args.steps, args.warmup, world, torch.cuda.synchronize, dist, dev)
        assert all(torch.isfinite(o).all() for o in outs)
    if n_str > 1:
        dt, outs = timed_region(one_step_streams, args.steps, max(args.warmup, n_str), world, torch.cuda.synchronize, dist, dev, keep=n_str)   # every lane warmed
        torch.cuda.synchronize()
        assert all(torch.isfinite(o).all() for o in outs)                  # the last step of EVERY lane
    else:
        dt = dt1
    imgs = world * Bz * args.steps
    line = {
        "metric": "images/sec at 15-step Natural Inference (CIFAR10 32x32, NCSN++)" + (f", {n_str}-stream throughput" if n_str > 1 else ""),
        "value": round(imgs / dt, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
    }
    if n_str > 1 and dt1 is not None:
        line["single_stream"] = {"value": round(imgs / dt1, 2), "ms_per_step": round(dt1 / args.steps * 1e3, 3)}
    if rank == 0 and n_str > 1:
        # the same K batches through the PRODUCTION pipeline (generate_sharded: Philox noise by global index drawn on the lane's stream, 15 steps, to_pixel, uint8
        # images to the host once at the end) -- what the reference's per-batch loop contains beyond the timed region above (:290 noise, :308-309 to_pixel + copy)
        from naturaldiffusion_amd.CIFAR10NaturalInference import generate_sharded
        pl = [e_ for e_, _, _ in lanes]
        generate_sharded(pl, None, 2 * Bz, Bz, device=dev, coeff=(C, Bm, node))                   # slabs of the lanes' first use
        torch.cuda.synchronize(); tp = time.perf_counter()
        im, _ = generate_sharded(pl, None, args.steps * Bz, Bz, device=dev, coeff=(C, Bm, node), to_cpu=True)
        assert int(im.shape[0]) == args.steps * Bz
        line["pipeline"] = {"value": round(args.steps * Bz / (time.perf_counter() - tp), 2)}
    line["config"] = {"workload": f"CIFAR10 NI 15-step {os.path.basename(args.weights)} B={Bz}/GPU NCSN++ 61.8M bf16, ni_step fp64 history",
                      "nfe": n_step, "batch_per_gpu": Bz, "streams": n_str, "sharding": f"batch x{world}, no collective"}

    if rank == 0 and not args.no_roofline:
        # ---- instrumented replica of the timed region: HIP events around every engine launch group and every ni_step launch (same stream)
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
        for i in range(args.steps):
            one_step(i)
        torch.cuda.synchronize()
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
        cg_flops = sum(fl(r) for r in rows if r[6].startswith("conv_gn") and not is8(r)) * fwd
        c8_flops = sum(fl(r) for r in rows if r[6].startswith("conv_gn") and is8(r)) * fwd
        gemm_flops = GFLOP_PER_IMAGE_FORWARD * 1e9 * Bz * fwd - cg_flops - c8_flops          # the rest: the k_gemm_* launches, the fused attention, the head
        all_ms = cg_ms + c8_ms + gemm_ms + other_ms
        ach = cg_flops / (cg_ms * 1e-3) / 1e12
        line["roofline"] = {"kernel": "k_conv_gn2+k_conv_gn3<32|16>", "bound": "mfma", "achieved": r4(ach), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": r4(ach / MFMA_BF16_PEAK_TFLOPS), "traffic": profiled_traffic("k_conv_gn", exclude=("k_conv_gn2<8", "k_conv_gn2<4")),
                            "launches": int(cg_n), "mean_launch_ms": r4(cg_ms / cg_n), "flops_per_launch": r4(cg_flops / cg_n), "share": r4(cg_ms / all_ms)}
        if c8_n:
            ach8 = c8_flops / (c8_ms * 1e-3) / 1e12
            line["roofline_conv_gn8"] = {"kernel": "k_conv_gn2<8|4>", "achieved": r4(ach8), "frac": r4(ach8 / MFMA_BF16_PEAK_TFLOPS), "launches": int(c8_n),
                                         "mean_launch_ms": r4(c8_ms / c8_n), "share": r4(c8_ms / all_ms)}
        ach_g = gemm_flops / (gemm_ms * 1e-3) / 1e12
        line["roofline_gemm"] = {"kernel": "k_gemm_*+k_attn_blk256+k_head_conv", "achieved": r4(ach_g), "frac": r4(ach_g / MFMA_BF16_PEAK_TFLOPS),
                                 "launches": int(gemm_n), "mean_launch_ms": r4(gemm_ms / gemm_n), "share": r4(gemm_ms / all_ms),
                                 "other_share": r4(other_ms / all_ms)}
        whole = (cg_flops + c8_flops + gemm_flops) / (all_ms * 1e-3) / 1e12
        line["roofline_whole_denoiser"] = {"achieved": r4(whole), "frac": r4(whole / MFMA_BF16_PEAK_TFLOPS)}
        bpe = ni_step_bytes_per_element(C)
        tot_ms = sum(a0.elapsed_time(a1) for _, a0, a1 in ev)
        tot_bytes = sum(bpe[k] * E for k, _, _ in ev)
        ach_gbs = tot_bytes / (tot_ms * 1e-3) / 1e9
        line["roofline_ni_step"] = {"kernel": "k_step_f64hist", "bound": "hbm", "achieved": r4(ach_gbs), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": r4(ach_gbs / HBM_PEAK_GBS), "traffic": profiled_traffic("k_step_f64hist"),
                                    "mean_launch_ms": r4(tot_ms / len(ev)), "bytes_per_launch": r4(tot_bytes / len(ev))}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (eager-PyTorch restatement of the reference path, fp32 NCSN++ +
        # ---- fp64 recurrence) on this host's cores; two bounded points
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
            points.append({"images": nb, "nfe": int(nodec.shape[0] - 1), "seconds": dc, "images_per_s": nb / dc, "denoiser_s": spent["t"]})
            if nb == 64:
                final64, z64 = xs[-1], z
        mp = points[-1]
        line["cpu_baseline"] = {"value": r4(mp["images_per_s"]), "unit": "images/s", "cores": threads, "kind": "port",
                                "sample": f"{mp['images']} img x {mp['nfe']} steps oracle fp32+fp64 {mp['seconds']:.1f}s (denoiser {mp['denoiser_s']:.1f}s)",
                                "physical_cores": physical, "config1": r4(points[0]["images_per_s"])}
        # ---- accuracy of the bf16 engine at the image level (stand-in for the FID delta, which is blocked on assets): the same
        # ---- 64 noise tensors through the HIP path, against the fp32 / fp64 oracle trajectory just computed
        if final64 is not None and os.path.basename(args.weights) == "step_15_weight_173.npz":
            eng64 = NCSNppEngine(synthetic_flat_params(0), max_batch=64, device=dev)
            got = CifarNI(C, Bm, node, 64 * 3 * 32 * 32, device=dev).run(eng64, z64.to(dev)).cpu()
            pg, pr = O.to_pixel(got).to(torch.int16), O.to_pixel(final64).to(torch.int16)
            rms = lambda t: float((t.double() ** 2).mean().sqrt())
            line["accuracy"] = {"rel_rms": r4(rms(got - final64) / rms(final64)), "uint8_mean_abs_diff": r4(float((pg - pr).abs().float().mean())),
                                "fid": "blocked"}
    del engine, ni, lanes
    return line


def sd3_reduced_depth_accuracy(dev, weights_csv="sd3_step_28_weight_sharp.csv", n=2, seed=3):
    """What fp8 operands cost over a whole 28-step SD3-form Natural Inference run, at a depth the fp32 CPU oracle finishes in seconds:
    a 4-block / 256-wide MMDiT (4 heads x 64, 16x16 image tokens + 29 text tokens, 16-channel 32x32 latents), CFG 7, the shipped
    coefficient file -- final latents of the HIP engine with bf16 operands and with NATINF_MMDIT_FP8 against oracle/mmdit_oracle.py
    (fp32; PARITY UNPINNED) + the oracle's restatement of the loop (src/SD3NaturalInference.py:198-223) on identical noise.
    Checker code: only bench.py's accuracy leg and tests/ call this."""
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
    return out


def bench_sd3(args, world, rank, dev, fp8=False, steps=2, warmup=1, flat=None):
    """BASELINE config 4 / 5 (see the module docstring)."""
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
    # this rank's images of a world * n image job sharded by global index (SD3NaturalInference.sd_generate_sharded's rule): Philox noise keyed by the index
    from naturaldiffusion_amd.SD3NaturalInference import philox_noise_f16
    from naturaldiffusion_amd.shard import rank_indices
    noises = philox_noise_f16(rank_indices(world * n, rank, world), (16, 128, 128), 10, dev)
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

    dt, outs = timed_region(lambda i: one_step(), steps, warmup, world, torch.cuda.synchronize, dist, dev)
    assert torch.isfinite(outs[-1].float()).all()
    D, L, tx = 1536, 24, 4096
    T = tx + tc
    flops_fwd_seq = L * (2.0 * T * 3 * D * D + 4.0 * T * T * D + 2.0 * T * D * D + 2.0 * T * 8 * D * D) - 2.0 * tc * 9 * D * D + 2.0 * tc * 4096 * D
    tf = flops_fwd_seq * 2 * n * nstep * steps / dt / 1e12
    line = {"metric": "images/sec at 28-step Natural Inference (SD3 1024x1024, MMDiT)", "value": round(world * n * steps / dt, 4),
            "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8+bf16" if fp8 else "bf16", "data": "synthetic",
            "frac": r4(tf / MFMA_BF16_PEAK_TFLOPS),
            "config": {"workload": f"SD3 NI 28-step {wname} 1024x1024, 4 img x CFG/GPU = 8 seq x (4096+333) tokens, MMDiT 2.03B " + ("fp8 e4m3 + bf16" if fp8 else "bf16"),
                       "nfe": 2 * nstep, "images_per_gpu": n, "sharding": f"batch x{world}, no collective", "streams": "image + text stream of one forward on two HIP streams"},
            "roofline": {"kernel": "whole forward", "bound": "mfma", "achieved": r4(tf), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": r4(tf / MFMA_BF16_PEAK_TFLOPS), "traffic": None}}
    if rank == 0 and not args.no_roofline:
        # ---- per-kernel rooflines, measured live with HIP events on the kernels' own ABI entry points at the engine's shapes
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
        # the same kernels where they actually run: one more 28-step batch with an event pair around every k_flash_attn64 launch AND every matmul-shaped launch
        # of the engine (natinf_gemm_profile: the kernel with the epilogue the engine gives it, on its stream, between its real neighbours)
        import ctypes
        check(lib.natinf_attention_profile(1), "attention_profile")
        check(lib.natinf_gemm_profile(1), "gemm_profile")
        one_step(); torch.cuda.synchronize()
        ms_tot, n_l = ctypes.c_double(), ctypes.c_int64()
        check(lib.natinf_attention_profile_read(ctypes.byref(ms_tot), ctypes.byref(n_l)), "attention_profile_read")
        check(lib.natinf_attention_profile(0), "attention_profile")
        check(lib.natinf_gemm_profile(0), "gemm_profile")
        gbuf = ctypes.create_string_buffer(1 << 16)
        check(min(0, lib.natinf_gemm_profile_read(gbuf, len(gbuf))), "gemm_profile_read")
        t_fa = ms_tot.value / max(1, n_l.value) * 1e-3
        fa_flops = 4.0 * T * T * 64 * H * Bs
        line["roofline"] = {"kernel": "k_flash_attn64_v2<1,64>", "bound": "mfma", "achieved": r4(fa_flops / t_fa / 1e12), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": r4(fa_flops / t_fa / 1e12 / MFMA_BF16_PEAK_TFLOPS), "traffic": None, "mean_launch_ms": r4(t_fa * 1e3), "launches": int(n_l.value),
                            "flops_per_launch": r4(fa_flops), "iso_ms": r4(t_fa_iso * 1e3), "flop_share": r4(L * fa_flops / (flops_fwd_seq * Bs))}
        # image-stream projections (4,096 tokens x 8 sequences per launch) by role; a row of the table: "M N K K1 taps batch kernel/eEPI launches total_ms"
        role = {(tx, 2 * D, D): "qk", (D, tx, D): "vT", (tx, D, D): "out", (tx, 4 * D, D): "fc1", (tx, D, 4 * D): "fc2"}
        tot_t, tot_f, tot_ideal, per, kernels = 0.0, 0.0, 0.0, {}, set()
        for row in gbuf.value.decode().splitlines():
            f = row.split()
            M_, N_, K_, bt, kern, nl, ms = int(f[0]), int(f[1]), int(f[2]), int(f[5]), f[6], int(f[7]), float(f[8])
            if bt == 1 and M_ == Bs * tx:
                M_, bt = tx, Bs                                               # (a flat launch over all sequences)
            name = role.get((M_, N_, K_))
            if name is None or bt != Bs:
                continue                                                      # text stream (333 tokens), embedders, modulation, proj_out: < 4 % of the forward's flops
            pk_ = MFMA_FP8_PEAK_TFLOPS if "fp8" in kern else MFMA_BF16_PEAK_TFLOPS
            f_, t_ = 2.0 * M_ * N_ * K_ * bt, ms / nl * 1e-3
            per[name] = [r4(f_ / t_ / 1e12), r4(f_ / t_ / 1e12 / pk_), r4(t_ * 1e6)]      # [TFLOP/s, fraction of ITS operand type's dense peak, mean launch us] -- in the engine
            kernels.add(kern)
            tot_t += t_; tot_f += f_; tot_ideal += f_ / (pk_ * 1e12)
        by_name = {}
        for kn in sorted(kernels):                                            # "w128_256x256/e1" ... -> "w128_256x256/e1+4+7+8" (the line stays short)
            nm, ep = kn.split("/e")
            by_name.setdefault(nm, []).append(ep)
        line["roofline_gemm"] = {"kernel": ",".join(f"{nm}/e{'+'.join(ep)}" for nm, ep in by_name.items()), "where": "in-engine (HIP events around every launch of one 28-step batch)",
                                 "achieved": r4(tot_f / tot_t / 1e12), "peak": r4(tot_f / tot_ideal / 1e12), "frac": r4(tot_ideal / tot_t), "shapes": per}
        # isolated back-to-back loop of fc1 WITH the engine's epilogue (tanh-GELU; fp8: + e4m3 output with E8M0 block scales), for the isolated-vs-engine gap
        M = Bs * tx
        N_, K_ = 4 * D, D
        if fp8:
            a8 = torch.randint(0, 255, (M, K_), device=dev, dtype=torch.uint8); b8 = torch.randint(0, 255, (N_, K_), device=dev, dtype=torch.uint8)
            a8 &= 0x77; b8 &= 0x77                                                     # finite e4m3 patterns
            sa, sb = torch.ones(M, device=dev), torch.ones(N_, device=dev)
            c8 = torch.empty(M, N_, device=dev, dtype=torch.uint8); cmx = torch.empty(N_ // 32, (M + 255) // 256 * 256, device=dev, dtype=torch.uint8)
            t_ = timed(lambda: check(lib.natinf_debug_gemm_fp8(M, N_, K_, ptr(a8), ptr(sa), None, ptr(b8), ptr(sb), None, ptr(c8), ptr(cmx), 3 | (2 << 8), 1, stream_ptr()), "gemm_fp8"))
        else:
            a = torch.randn(M, K_, device=dev).bfloat16(); b = torch.randn(N_, K_, device=dev).bfloat16()
            c = torch.empty(M, N_, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N_, device=dev)
            t_ = timed(lambda: check(lib.natinf_debug_gemm_fused(0, M, N_, K_, ptr(a), ptr(b), ptr(bias), None, None, None, 30, None, None, 1.0, 2, ptr(c), 0, None, None, 0,
                                                                  stream_ptr()), "gemm_fused"))
        line["roofline_gemm"]["iso_fc1"] = r4(2.0 * M * N_ * K_ / t_ / 1e12)
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
        line["cpu_baseline"] = {"value": r4(1.0 / per_image_s), "unit": "images/s", "cores": threads, "kind": "port",
                                "sample": f"mmdit oracle fp32: 1 seq x 2 of 24 blocks {d2:.1f}s, extrapolated to {per_image_s:.0f}s/image"}
        del eng
        acc = sd3_reduced_depth_accuracy(dev, wname)
        line["acc"] = {"bf16_rel_rms": acc["bf16"]["rel_rms"], "fp8_rel_rms": acc["fp8"]["rel_rms"], "pinned": False}
        eng = None
    del eng, ni
    return line


def bench_fid50k(args, world, rank, dev, steps=1, warmup=0):
    """BASELINE config 3 (see the module docstring): generate_sharded + Inception pool3 engine + calc_fid_sharded, both matrices."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from naturaldiffusion_amd import _lib, coeffgen
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    from naturaldiffusion_amd.coeff import load_coeff_npz
    from naturaldiffusion_amd.fid_stats import ActivationStats, frechet_distance
    from naturaldiffusion_amd.inception import InceptionEngine, load_fid_inception_weights
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine, load_score_sde_checkpoint
    from naturaldiffusion_amd.synth import synthetic_flat_params, synthetic_inception_flat
    _lib.require_gpu()
    total = int(args.fid_samples)
    share_of = world if world > 1 else (args.fid_share_of or 8)           # one GPU: rank 0's share of an 8-way job unless told otherwise
    vrank = rank if world > 1 else 0
    ckpt = ROOT / "deps/score_sde_pytorch/checkpoint_8.pth"
    ref_path = ROOT / "weights/cifar10_mu_sigma.npz"
    inc_path = M.inception_weights_path()
    blocked = [n for n, p in (("checkpoint_8.pth", ckpt), ("inception weights", inc_path), ("cifar10_mu_sigma.npz", ref_path)) if not Path(p).exists()]
    flat = load_score_sde_checkpoint(str(ckpt)) if ckpt.exists() else synthetic_flat_params(0)
    engine = NCSNppEngine(flat, max_batch=args.batch, device=dev)
    lanes = [engine] + [engine.clone() for _ in range(max(1, args.streams) - 1)]
    inception = InceptionEngine(load_fid_inception_weights(inc_path) if Path(inc_path).exists() else synthetic_inception_flat(0), max_batch=M.FID_BATCH, device=dev)
    from naturaldiffusion_amd.fid_stats import FrechetReference
    ref = (str(ref_path) if ref_path.exists() else FrechetReference(np.zeros(2048), np.eye(2048)))       # (one object for the whole job: the reference covariance's square root is taken once)
    mats = [("dpmsolverpp2s_018", load_coeff_npz(ROOT / "results/dpmsolverpp/dpmsolverpp2s_018.npz")),
            ("ddim_vp_018", coeffgen.ddim_vp_continuous(np.linspace(1.0, 1e-3, 19)))]
    for e_ in lanes:                                                       # set-up, not a step: code objects load, workspaces are touched
        e_(torch.zeros(2, 3, 32, 32, device=dev), torch.full((2,), 500.0, device=dev))
    inception(torch.zeros(2, 32, 32, 3, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    acc = {"gen": 0.0, "inception": 0.0, "allreduce": 0.0, "frechet": 0.0}
    res = {}

    def one_step(i):
        pend = []
        for name, coeff in mats:
            t0 = time.perf_counter()
            imgs, idx = M.generate_sharded(lanes, None, total, args.batch, rank=vrank, world=share_of, seed=888, device=dev, to_cpu=False, coeff=coeff)
            torch.cuda.synchronize()
            acc["gen"] += time.perf_counter() - t0
            tm = {}
            # rank 0 holds the job's FID (the others wait in the closing barrier); its host part runs on a thread while the next matrix's images are generated
            fid = M.calc_fid_sharded(imgs, ref, dev, model=inception, timings=tm, root_only=True, defer=True)
            for k in ("inception", "allreduce"):
                acc[k] += tm[k + "_s"]
            pend.append((name, fid))
            res[name] = {"fid": None, "images_local": int(imgs.shape[0]), "images_all": tm["images_all_ranks"], "imgs": imgs}
        t0 = time.perf_counter()
        for name, fid in pend:
            res[name]["fid"] = fid.result()
            acc["frechet"] += fid.seconds                                   # the threads' own time ...
        acc["frechet_wait"] = acc.get("frechet_wait", 0.0) + time.perf_counter() - t0      # ... and what of it the step still waits for
        return res

    dt, _ = timed_region(one_step, steps, warmup, world, torch.cuda.synchronize, dist, dev)
    n_local = res[mats[0][0]]["images_local"]
    n_all = res[mats[0][0]]["images_all"]
    runs = steps + warmup
    per = {k: v / runs for k, v in acc.items()}                           # per pass over both matrices
    line = {"metric": "images/sec generated and FID-scored (CIFAR10 50k-image FID job, DPM-Solver++(2S) and DDIM coefficient matrices, 18 NFE)",
            "value": round(len(mats) * n_all * steps / dt, 2), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 1), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "images": n_local, "share_of": share_of, "batches": int(np.ceil(n_local / args.batch)), "last_batch": n_local - (int(np.ceil(n_local / args.batch)) - 1) * args.batch,
            "s": {k: r4(v) for k, v in per.items()}, "gen_rate": r4(len(mats) * n_local / per["gen"]), "inc_rate": r4(len(mats) * n_local / per["inception"]),
            "fid": ("blocked" if blocked else {k: r4(v["fid"]) for k, v in res.items()}),
            "config": {"workload": f"CIFAR10 FID job {total} images / {share_of} shares, dpmsolverpp2s_018.npz + coeffgen.ddim_vp_continuous(19 nodes), B={args.batch}, "
                                   "Inception pool3 engine in 500s, one statistics all-reduce", "nfe": 18, "streams": len(lanes), "sharding": f"batch x{share_of}"}}
    if dist.is_available() and dist.is_initialized():
        line["collective"] = f"{dist.get_backend()} x{dist.get_world_size()}"        # what carried s.allreduce (absent: no process group, nothing to reduce over)
    if blocked:
        line["frechet_vs_synthetic_ref"] = {k: r4(v["fid"]) for k, v in res.items()}
    # untimed sanity figure that needs no asset: the two matrices integrate the same ODE from the same noise with the same network, so their image
    # statistics must be close (statistics of THIS rank's images)
    if rank == 0:
        st = []
        for name, _ in mats:
            s_ = ActivationStats(2048, device=dev)
            im = res[name]["imgs"]
            for i in range(0, len(im), 50):
                s_.update(inception(im[i:i + 50]))
            st.append(s_.mean_cov())
        line["d_matrices"] = r4(frechet_distance(st[0][0], st[0][1], st[1][0], st[1][1]))
        inc_tf = INCEPTION_GFLOP_PER_IMAGE * 1e9 * len(mats) * n_local / per["inception"] / 1e12
        line["roofline"] = {"kernel": "inception pool3 engine", "bound": "mfma", "achieved": r4(inc_tf), "peak": MFMA_BF16_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": r4(inc_tf / MFMA_BF16_PEAK_TFLOPS), "traffic": None}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "fid50k":
        # bounded CPU sample of the same job: 16 images x 18 steps through the oracle denoiser + recurrence, then the Inception oracle on them
        from oracle import ni_oracle as O, ncsnpp_oracle as N, inception_oracle as IO
        from naturaldiffusion_amd.synth import synthetic_state_dict
        threads = max(1, min(32, torch.get_num_threads()))
        torch.set_num_threads(threads)
        model = N.model_fn_from_params(synthetic_state_dict(0))
        z = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(888))
        t0 = time.perf_counter()
        x = O.cifar_ni_trajectory(model, z, *mats[0][1])[-1]
        IO.forward(IO.make_params(0), O.to_pixel((x + 1) / 2).permute(0, 3, 1, 2).float() / 255)
        dc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": r4(16 / dc), "unit": "images/s", "cores": threads, "kind": "port",
                                "sample": f"16 img x 18 steps oracle NCSN++ fp32 + fp64 recurrence + Inception oracle, {dc:.1f}s"}
    del engine, lanes, inception, res
    return line


def bench_validate(args, world, rank, dev, steps=3, warmup=1):
    """SURVEY 8f N2 + N4 (see the module docstring): ValidateNaturalInference.natural_inference("ddim", 24) on the DiT-XL/2 engine + VAE decode."""
    import tempfile
    import torch
    import torch.distributed as dist
    from naturaldiffusion_amd import _lib
    from naturaldiffusion_amd import ValidateNaturalInference as V
    from naturaldiffusion_amd.dit import DiTEngine, flatten_state_dict, XL2
    from naturaldiffusion_amd.synth import synthetic_dit_state_dict, synthetic_vae_flat
    from naturaldiffusion_amd.vae import VAEDecoder
    _lib.require_gpu()
    n, nstep = 8, 24
    V.device = str(dev)
    dit = DiTEngine(flatten_state_dict(synthetic_dit_state_dict(XL2["depth"], XL2["hidden"], seed=0), XL2["depth"], XL2["hidden"]), 2 * n, device=dev, **XL2)
    vae = VAEDecoder(synthetic_vae_flat(4), max_batch=n, latent_ch=4, latent_res=32, device=dev)
    outdir = Path(tempfile.mkdtemp(prefix="natinf_validate_"))
    t_acc = {"dit": 0.0, "dit_n": 0, "vae": 0.0, "vae_n": 0}
    ev = lambda: torch.cuda.Event(enable_timing=True)

    class TimedDiT:
        max_batch = 2 * n                                                   # the CFG pair of a step rides in ONE forward of 16 (ValidateNaturalInference._cond_uncond)

        def forward(self, z, t, y):
            a, b = ev(), ev()
            a.record(); out = dit(z, t, y); b.record()
            t_acc.setdefault("ev", []).append((a, b))
            return out

    def decode(latents, path):
        a, b = ev(), ev()
        a.record(); img = vae(latents); b.record()
        t_acc.setdefault("evv", []).append((a, b))
        V.save_image_grid(img, outdir / Path(path).name)
        return img
    V.denoiser_factory, V.decoder_factory = (lambda: TimedDiT()), (lambda: decode)
    try:
        dt, outs = timed_region(lambda i: V.natural_inference("ddim", nstep), steps, warmup, world, torch.cuda.synchronize, dist, dev)
    finally:
        V.denoiser_factory = V.decoder_factory = None
    assert torch.isfinite(outs[-1]).all()
    dit_ms = sum(a.elapsed_time(b) for a, b in t_acc["ev"]) / len(t_acc["ev"])
    vae_ms = sum(a.elapsed_time(b) for a, b in t_acc["evv"]) / len(t_acc["evv"])
    dit_fl = XL2["depth"] * 256 * (24 * XL2["hidden"] ** 2 + 4 * 256 * XL2["hidden"]) * 2 * n        # 2*MAC per forward of 16 samples (tools/bench_dit.py)
    line = {"metric": "images/sec at 24-step Natural Inference (DiT-XL/2 256x256, CFG 4, DDIM matrix) incl. VAE decode",
            "value": round(world * n * steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "dit_ms": r4(dit_ms), "dit_forwards": len(t_acc["ev"]) // (steps + warmup), "vae_ms": r4(vae_ms),
            "config": {"workload": "ValidateNaturalInference.natural_inference('ddim', 24): DiT-XL/2 engine, CFG 4 (cond + uncond as one forward of 16), natinf_step_f32prod, "
                                   "AutoencoderKL decoder engine 8 x 256x256 + PNG row", "nfe": 2 * nstep, "images_per_gpu": n},
            "roofline": {"kernel": "DiT-XL/2 forward B=16", "bound": "mfma", "achieved": r4(dit_fl / (dit_ms * 1e-3) / 1e12),
                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": r4(dit_fl / (dit_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS), "traffic": None}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "validate":
        from oracle import dit_oracle as DO
        threads = max(1, min(32, torch.get_num_threads()))
        torch.set_num_threads(threads)
        P = DO.make_params(4, XL2["hidden"], seed=3)                       # 4 of 28 blocks at full width, extrapolated
        g = torch.Generator().manual_seed(5)
        x, t, y = torch.randn(n, 4, 32, 32, generator=g), torch.full((n,), 500.0), torch.zeros(n, dtype=torch.long)
        t0 = time.perf_counter()
        DO.forward(P, x, t, y, XL2["heads"])
        d4 = time.perf_counter() - t0
        per_batch = d4 / 4 * XL2["depth"] * 2 * nstep
        line["cpu_baseline"] = {"value": r4(n / per_batch), "unit": "images/s", "cores": threads, "kind": "port",
                                "sample": f"dit oracle fp32: B=8 through 4 of 28 blocks {d4:.1f}s, extrapolated to {per_batch:.0f}s per 8 images (no VAE)"}
    del dit, vae
    return line


if __name__ == "__main__":
    main()
