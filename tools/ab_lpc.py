"""A/B of the deterministic block kernel's lanes-per-cell shape for p = 2, 5, 8 (tools only): the library as built (cells inside one
wave: 16 / 64 / 128 lanes per cell) against an experimental build with n^2 lanes per cell (9 / 36 / 81: cells span waves, the tile
exchanges go through the workgroup barrier, 28 / 7 / 3 cells per pass, hardly any idle lane), and the atomic pencil kernel that bench.py
still prefers there.  Every case is one `bench.py` child process (the library is chosen per process through BP5_LIB).

  python tools/ab_lpc.py [--exp deal-and-ceed-on-gpu_amd/libbp5_exp.so] [--degrees 2 5 8]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--exp", default=os.path.join(ROOT, "deal-and-ceed-on-gpu_amd", "libbp5_exp.so"))
ap.add_argument("--degrees", type=int, nargs="+", default=[2, 5, 8])
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
CASES = {2: [("default", (8, 8, 4), 56), ("exp", (8, 14, 2), 56), ("exp", (8, 8, 4), 56), ("default", (0, 0, 0), 0)],
         5: [("default", (4, 4, 2), 56), ("exp", (4, 4, 2), 56), ("exp", (6, 4, 2), 56), ("default", (8, 8, 8), 0)],
         8: [("default", (2, 2, 2), 56), ("exp", (2, 2, 2), 56), ("exp", (4, 2, 2), 56), ("default", (8, 8, 8), 0)]}
for p in args.degrees:
    for lib, block, variant in CASES[p]:
        env = dict(os.environ)
        if lib == "exp":
            env["BP5_LIB"] = args.exp
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--degree", str(p), "--steps", str(args.steps), "--warmup", "3", "--no-cpu-baseline",
               "--no-traffic-pass", "--sustained-iters", "0", "--apply-variant", str(variant), "--cell-block", *[str(b) for b in block]]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"p={p} {lib} block={block} variant={variant}: FAILED\n{r.stderr[-800:]}", flush=True)
            continue
        d = json.loads(line[0])
        print(f"p={p} lib={lib:7s} block={block} variant={variant:2d}: {d['value'] / 1e9:6.2f} GDoF/s, {d['ms_per_step']:.3f} ms/iter, CG frac {d['roofline_cg']['frac_of_hbm_peak']:.3f}, "
              f"kernel {d['roofline']['kernel']} {d['roofline']['avg_launch_ms']:.3f} ms, fused {d['config']['cg_dot_products_fused']}", flush=True)
