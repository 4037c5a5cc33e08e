"""Per-frame latency of the stateful entry points (host frame in -> flow vectors out), the call the ROS node makes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_optic_flow_amd import FftMethod, FastSpacedBMMethod, ScaleRotationEstimator, synth

def timeit(fn, n=300):
    for _ in range(20): fn()
    t = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t) / n * 1e6

frames = [synth.pair_np(k, 480, 480, 3, -2)[0] for k in range(4)]
fm = FftMethod(480, 120, 80.0); i = [0]
def f():
    fm.processImage(frames[i[0] & 3]); i[0] += 1
print(f"FftMethod(480,120).processImage            : {timeit(f):8.1f} us/frame")
def g():
    fm.processImageLongRange(frames[i[0] & 3]); i[0] += 1
print(f"FftMethod(480,120).processImageLongRange   : {timeit(g):8.1f} us/frame")
fm64 = FftMethod(448, 64, 80.0); fr64 = [f_[:448, :448].copy() for f_ in frames]
def h():
    fm64.processImage(fr64[i[0] & 3]); i[0] += 1
print(f"FftMethod(448,64).processImage             : {timeit(h):8.1f} us/frame")
wide = [synth.pair_np(k, 480, 752, 3, -2)[0] for k in range(4)]
bm = FastSpacedBMMethod(16, 16, 8, (480, 752))
def b():
    bm.processImage(wide[i[0] & 3]); i[0] += 1
print(f"FastSpacedBMMethod(16,16,8) 752x480        : {timeit(b):8.1f} us/frame")
sr = ScaleRotationEstimator(480, 49.9)
def s():
    sr.processImage(frames[i[0] & 3]); i[0] += 1
print(f"scaleRotationEstimator(480).processImage   : {timeit(s, 100):8.1f} us/frame")
