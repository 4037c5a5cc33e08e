import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, synth
n, grid, origin, stride, (h, w), k, const = 124, (1, 2), (7, 8), (136, 123), (256, 136), 988, (0, 39)
video, _ = synth.video_torch(2, h, w, "cpu", k=k)
video[const[0]] = const[1]
fr = video.numpy()
fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=grid, origin=origin, stride=stride)
dv = video.cuda()
got = fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy()[0]
lay = O.fft_layout(w, h, n, grid[0], grid[1], origin, stride)
w64, _ = O.fft_process(fr[1], fr[0], lay, 64); w32, _ = O.fft_process(fr[1], fr[0], lay, 32)
print("got", got, "\nw64", w64, "\nw32", w32, "\nerr64", np.abs(got - w64), "o32-o64", np.abs(w32 - w64))
