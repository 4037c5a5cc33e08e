"""Random block-matching geometries (blocks 4..128, radii 1..48, any step / grid) through the GPU path against the oracle,
bit for bit. usage (GPU box): python tools/bm_fuzz.py [seed] [trials]"""
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import oracle_lib as O
from mrs_optic_flow_amd import FastSpacedBMMethod
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    block = int(rng.choice([4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 52, 64, 72, 96, 100, 120, 124, 128]))
    radius = int(rng.integers(1, 49))
    step = int(rng.choice([0, 1, 3, 4, 8, 24]))
    if (block + 2 * radius) * (block // 4 + (2 * radius + 4) // 4 + 9) * 4 > 150 * 1024:
        continue
    gx, gy = int(rng.integers(1, 7)), int(rng.integers(1, 4))
    S = block + step
    w = gx * S + 2 * radius + int(rng.integers(0, S)); h = gy * S + 2 * radius + int(rng.integers(0, S))
    cur = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    prev = np.roll(cur, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(1, 2))
    prev = np.clip(prev.astype(np.int32) + rng.integers(-6, 7, prev.shape), 0, 255).astype(np.uint8)
    try:
        eng = FastSpacedBMMethod(block, radius, step, (h, w))
    except Exception as e:
        print("unsupported", block, radius, step, str(e)[:60]); continue
    cfg = O.bm_config_fast_spaced(w, h, block, step, radius)
    dx, dy, mode = eng.process_batch_host(cur, prev)
    for k in range(2):
        wdx, wdy, wmode = O.bm_process(cur[k], prev[k], cfg)
        if not ((dx[k] == wdx).all() and (dy[k] == wdy).all() and tuple(mode[k, :2]) == wmode):
            bad += 1; print("MISMATCH", trial, block, step, radius, w, h)
print("done, mismatches:", bad)
