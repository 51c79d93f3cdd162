"""Soak: natural_inference_tx at the bench's shape (batches of 512, 15 steps), one stream against two, several repetitions: bit-identical images?"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd import CIFAR10NaturalInference as M
from naturaldiffusion_amd.synth import synthetic_flat_params
flat = synthetic_flat_params(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
run = lambda s: M.natural_inference_tx(batch_size=512, weight_path="/root/repo/weights/step_15_weight_173.npz", sample_count=512 * nb, seed=3, device="cuda:0",
                                       compute_fid=False, flat_params=flat, streams=s)
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    a = run(1)
bad = 0
for rep in range(4):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        b = run(2)
    dt = time.perf_counter() - t0
    same = torch.equal(a, b)
    bad += int(not same)
    print(f"rep {rep}: {512 * nb} images on two streams in {dt:.2f} s (set-up included); identical to one stream: {same}", flush=True)
print("mismatching repetitions:", bad)
