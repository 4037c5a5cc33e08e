"""Scratch GPU probe: smoke + quick timing of K1/K2 on synthetic batches (not the bench contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as g
from mrs_optic_flow_amd import FftMethod, FastSpacedBMMethod, synth

g.smoke()
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
for (H, W, N, grid, origin, stride, B) in [(480, 752, 64, (8, 8), (1, 1), (98, 59), 256), (1080, 1920, 128, (16, 16), (0, 0), (119, 63), 32)]:
    cur, prev, shifts, kinds = synth.batch_torch(B, H, W, N // 8, dev)
    fm = FftMethod(sample_point_size=N, frame_shape=(H, W), grid=grid, origin=origin, stride=stride)
    out = fm.process_batch_device(cur, prev); torch.cuda.synchronize()
    med = np.nanmedian(out.cpu().numpy(), axis=1)
    err = np.abs(med - shifts.numpy()).max()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): fm.process_batch_device(cur, prev, out=out)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    print(f"FFT N={N} {W}x{H} B={B}: {ms:.3f} ms/launch, {B/ms*1e3:.0f} pairs/s, median-shift err {err:.3f}")
H, W, B = 480, 752, 256
cur, prev, shifts, kinds = synth.batch_torch(B, H, W, 12, dev)
bm = FastSpacedBMMethod(16, 16, 8, (H, W))
dx, dy, mode = bm.process_batch_device(cur, prev); torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5): bm.process_batch_device(cur, prev)
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 5
m = mode[:, :2].cpu().numpy()
ok = sum(1 for k in range(B) if kinds[k] == "shift" and tuple(m[k]) == tuple(-shifts[k].numpy()))
print(f"BM c3 B={B}: {ms:.3f} ms/launch, {B/ms*1e3:.0f} pairs/s; mode == -planted for {ok}/{sum(1 for k in kinds if k=='shift')} shift pairs")
