"""write_png_rgb (band-parallel deflate) against PIL's encoder on the Validate image row (260 x 2066 RGB), host only."""
import sys, time, tempfile, os
from pathlib import Path
import numpy as np
from PIL import Image
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from naturaldiffusion_amd.ValidateNaturalInference import write_png_rgb
r = np.random.RandomState(0)
smooth = np.clip(np.cumsum(r.randn(260, 2066, 3) * 6, axis=1) + 128, 0, 255).astype(np.uint8)      # image-like rows
noise = r.randint(0, 256, (260, 2066, 3)).astype(np.uint8)
p = tempfile.mktemp(suffix=".png")
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for name, a in (("noise", noise), ("smooth", smooth)):
    for rep in range(4):
        t0 = time.perf_counter(); write_png_rgb(a, p); d1 = time.perf_counter() - t0
        ok = np.array_equal(np.array(Image.open(p).convert("RGB")), a)
        t0 = time.perf_counter(); Image.fromarray(a).save(p + ".pil.png", compress_level=1); d2 = time.perf_counter() - t0
        print(f"{name} rep {rep}: write_png_rgb {d1 * 1e3:.1f} ms ({os.path.getsize(p)} B, same pixels: {ok}), PIL level 1 {d2 * 1e3:.1f} ms ({os.path.getsize(p + '.pil.png')} B)", flush=True)
        time.sleep(0.2)
