#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

A "step" is one pass of the hot path over one HBM-resident batch of synthetic frame pairs
through the C ABI (mof_*_process_batch_device, ONE kernel launch per batch for the FFT path):

  c2 (default)  FftMethod, 752x480, 8x8 grid of 64x64 patches, 1024 pairs per GPU   <- BASELINE metric
  c3            FastSpacedBMMethod, 752x480, sps 16 / step 8 / radius 16, 1024 pairs per GPU
  c4            FftMethod, 1920x1080, 16x16 grid of 128x128 patches, 1024 pairs per GPU (8192 over 8 GPUs)
  c5            c2 + scaleRotationEstimator (log-polar + whole-frame phase correlation of the 480^2 centre crop)
  c2seq / c5seq the same two on a VIDEO (1025 consecutive frames): the sequence entry points, every frame transformed once

Multi-GPU (--gpus N under torch.distributed.run): frame pairs are independent, so every rank owns
its own shard of the batch (weak scaling, no data-path collective); each step ends with the one
RCCL all-gather of the per-pair flow vectors.

Prints ONE JSON line on rank 0. `roofline.achieved` = algorithmic bytes per launch / average kernel
duration measured with HIP events on the launch stream inside the timed region; `cpu_baseline` = the
CPU oracle (a port of the reference's useOCL=false path; the reference itself cannot be built here)
timed on this box's host cores over a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X vector fp32 peak (all-FMA; an add-dominated FFT can reach about half of it)
# which pipe the FFT arithmetic of K1 runs on (the MFMA form of the forward row pass was built and measured)
COMPUTE_PIPE_NOTE = "VALU fp32 (Stockham passes in LDS; MFMA row-DFT variant: see DESIGN.md section 4, K1/MFMA experiment)"


def binding_note(name: str, wl) -> str:
    """What actually bounds the dominant kernel of a workload (measured)."""
    if wl["kind"] == "bm":
        if wl["block"] == 16 and wl["radius"] in (8, 16) and not wl.get("block_method"):
            return ("integer VALU: v_qsad_pk_u16_u8 at 1 byte-difference per lane per cycle; the wave's instruction stream runs at "
                    "92 % of its issue time, 75-77 % of the instruction's ceiling; not HBM (DESIGN.md section 4, K2)")
        return ("integer VALU: generic block scan 97 % VALU-busy, 81 % of those cycles in v_qsad_pk_u16_u8 (the rest: one "
                "v_pk_mov per odd window pair, widening); small geometries (c1) are staging-latency-bound; not HBM")
    if wl["kind"] == "fftlr":
        return ("planned kernel (run-time plan), long-range front end: the quarter-resolution pixels of four columns from two 16-byte runs per tapped row (r06); "
                "one patch per pair: launch- and latency-bound, not HBM")
    if wl["kind"] == "fft+2dt":
        return (f"1/4-scale resize fused into K1's load (half of the frame rows are fetched) + one {wl['n']} x {wl['n']} patch per pair: "
                "0.10 ms per 1024 pairs, the one path here whose time is mostly frame fetch (FETCH_SIZE factor of this load path "
                "calibrated by the callr workload, profiles/traffic_callr.json)")
    if wl["kind"] == "fft+rt":
        return "K1 as in ref + the getRT tail (one wavefront per pair, fp64 recurrences: 0.35 ms per 1024 pairs); not HBM"
    if wl["kind"] == "fftseq":
        if wl["n"] == 64:
            return ("K1 sequence kernel: one real forward transform + one Hermitian inverse per frame and patch (1.0 instead of 1.5 "
                    "complex-transform units), four barriers per frame. profiles/r04_c2seq_sq_pmc.csv (kernel unchanged in r05) (per launch of 65,536 patch "
                    "frames): VALU issue 47.8 % (272.3 M wave-instructions x 2 cycles over 1024 SIMDs x 1.114 M cycles), LDS active "
                    "56.7 % of the CU cycles (12.6 % of it bank conflicts); the two add to 104 %: the pipes run one after the other, "
                    "not side by side; not HBM")
        if wl["n"] > 192:
            return ("large-patch pipeline on a video (r06): the tuned row transforms once per FRAME (Zh slot = frame x patches + patch), the column "
                    "kernel on slots q | q + patches, inverse rows, tail; through Zh / Dt in HBM scratch; the transforms are bound by their own LDS / "
                    "VALU work at two waves per SIMD (DESIGN.md section 4, large patches / K4-K8)")
        return ("half-tile kernel K1h in its video form (pc_half_kernel<CH, M, SEQ>, r05; also 128 x 128 since it beat pc_seq_half.hip there: c4seq 93 k -> "
                "112 k): a frame's half spectrum stays in the registers of the forward column pass's last stage, where the next pair's "
                "cross-power meets it -- one image transform per pair instead of two; runs of consecutive pairs per workgroup, the run length "
                "picked per launch so that the workgroups fill whole rounds of the resident slots; same bits as the half-tile pair form; VALU + "
                "LDS in series as K1h; not HBM (DESIGN.md section 4, K1h on a video)")
    if wl["kind"] == "fft+srseq":
        return ("c5 on a video: one Lanczos4 remap and one real row transform per frame (K5s), column pass walking pairs in "
                "time with the previous spectra in registers (K6s); DESIGN.md section 4 (sequence mode)")
    if wl["kind"] == "fft+sr":
        return ("scale/rotation pipeline K4-K8 takes 76 % of the step (profiles/r06_c5_kernel_stats.csv, r06_c5_sq_pmc.csv): log-polar gathers "
                "(v_dot4c taps on LDS-staged source boxes; Lanczos4 515 us: LDS 58 % busy, VALU 30 %, waits 44 %; cubic 382 us) 33 %, whole-frame "
                "transforms through Zh / Dt (K5s 431 us at 5.4 TB/s, K6s 555 us at 5.0 TB/s, K7 193 us at 4.8 TB/s: the minimum bytes of a rows -> "
                "columns -> rows structure in f32, at 76-86 % of the achievable copy rate) 43 %; K1 (637 us) as in c2; DESIGN.md section 4 (K4-K8, r05 block)")
    if wl["n"] == 120:
        return ("half-tile kernel K1h (pc_half_kernel.hip, r05): each image transformed on its own on a 60 x 136 complex tile, TWO workgroups of "
                "8 waves per CU, every phase on all waves, the untangle / pairing / cross-power fused into the passes, the previous spectrum in "
                "registers, the current image's pixels requested under the previous image's column pass. profiles/r06_ref_sq_pmc.csv (per launch of "
                "16,384 patch pairs): VALU issue 51.7 % (3589 wave-instructions per wave x 2 cycles), LDS active 51.5 % of the CU cycles (26 % of it "
                "bank conflicts; r06: line order 0 2 1 3 in the later row stages), waves parked at a wait 30.9 % of their cycles -- against the tuned one-workgroup kernel's 38.8 % + 37.0 %, waits 45.6 % (r05_ref_tuned_sq_pmc.csv, "
                "MOF_FFT_HALF=0: 1.10 M pairs/s); VALU + LDS = 104 %: the two pipes in series, as K1; not HBM")
    if wl["n"] == 128:
        return ("one persistent workgroup per CU (the tile fills the LDS): the LDS store path (ds_write_b64 = 6 cycles per "
                "wave-instruction, 7.5 tile stores per patch pair) and the 8-wave inverse passes; N = 128: VALU issue ~42 %, "
                "LDS 46 % busy, 6 % of it bank conflicts (profiles/r06_c4_sq_pmc.csv; the kernel is unchanged since r03); the two add to "
                "88 %: in series; not HBM -- DESIGN.md section 4 (K1 at N = 128)")
    if wl["n"] in (60, 72, 90, 96, 100) or 136 <= wl["n"] <= 192:
        return ("half-tile kernel K1h (pc_half_kernel.hip, r05; planned Stockham stages with compile-time radices, sources / sinks fused into "
                "the passes, skew shift per size). profiles/r06_{p60,l160}_sq_pmc.csv, r05_p96_sq_pmc.csv: p60 VALU issue 55 % + LDS 61 % (30 % of it conflicts), waits 29 %; p96 "
                "47 % + 49 % (25 %), waits 33 %; l160 36 % + 41 % (38 %), waits 36 % at one 10-wave workgroup per CU -- LDS bank conflicts of the "
                "generic lane maps and the series of the two pipes; not HBM (DESIGN.md section 4, K1h)")
    if wl["n"] != 64:
        return ("planned kernel (run-time radix plan, pc_kernel_generic.hip / pc_large_kernel.hip): the general path, not tuned -- "
                "DESIGN.md section 4 (size-generic kernels)")
    return ("four workgroups per CU (LDS capacity). profiles/r06_c2_sq_pmc.csv (per launch of 65,536 patch pairs; the kernel and its counters are unchanged since r04): VALU issue 53.7 % "
            "(370.3 M wave-instructions x 2 cycles over 1024 SIMDs x 1.346 M cycles; 1412 per wave and patch pair), LDS active 52.7 % "
            "of the CU cycles (9.7 % of it bank conflicts; two thirds of it the store path, 6 cycles per ds_write_b64), waves stalled "
            "on LDS issue 14 %; VALU + LDS = 106 %: the two pipes run one after the other, not side by side -- that serialisation, "
            "not either pipe and not HBM, is what binds (DESIGN.md section 4, K1)")


def _tile_workload(n: int):
    """t<n>: a frame of (480 // n * n)^2 pixels tiled by n x n patches -- the reference tiling at another sample_point_size (sweeps)."""
    g = max(1, 480 // n)
    fs = g * n
    return dict(kind="fft", h=fs, w=fs, n=n, grid=(g, g), origin=(0, 0), stride=(n, n), batch=512 if n > 128 else 1024, s=min(n // 8, 15),
                name=f"t{n}: FftMethod {fs}x{fs}, {g}x{g} grid of {n}x{n} patches (reference tiling), batch per GPU as stated",
                bytes_per_pair=2 * fs * fs + g * g * 8)


def fft_flops_per_pair(n: int, patches: int) -> float:
    """fp32 work of one frame pair on the FFT path, BASELINE.md section 3 convention: per patch pair one complex 2-D
    forward transform (the two real images ride one complex transform) and half of one for the Hermitian inverse,
    5 N^2 log2(N^2) flop per complex 2-D transform, plus ~22 flop per bin of the half cross-power spectrum."""
    import math
    fft2 = 5.0 * n * n * math.log2(n * n)
    return patches * (1.5 * fft2 + 22.0 * n * n / 2)


WORKLOADS = {
    "c2": dict(kind="fft", h=480, w=752, n=64, grid=(8, 8), origin=(1, 1), stride=(98, 59), batch=1024, s=8,
               name="c2: FftMethod 752x480, 8x8 grid of 64x64 patches, batch=1024 frame pairs per GPU",
               # min(2*Gx*Gy*N^2, 2*W*H) u8 in + Gx*Gy*2*4 B out
               bytes_per_pair=min(2 * 64 * 64 * 64, 2 * 752 * 480) + 64 * 8),
    "c4": dict(kind="fft", h=1080, w=1920, n=128, grid=(16, 16), origin=(0, 0), stride=(119, 63), batch=1024, s=16,
               name="c4: FftMethod 1920x1080, 16x16 grid of 128x128 patches, batch=1024 frame pairs per GPU",
               bytes_per_pair=min(2 * 256 * 128 * 128, 2 * 1920 * 1080) + 256 * 8),
    # the reference's own default geometry (config/default.yaml:31-32): 480x480 crop, 4x4 patches of 120x120
    "ref": dict(kind="fft", h=480, w=480, n=120, grid=(4, 4), origin=(0, 0), stride=(120, 120), batch=1024, s=15,
                name="ref: FftMethod 480x480, 4x4 grid of 120x120 patches (reference default.yaml), batch=1024 per GPU",
                bytes_per_pair=2 * 480 * 480 + 16 * 8),
    # sizes WITHOUT a tuned kernel (r04): the run-time planned kernel (pc_kernel_generic.hip) and, beyond one CU's LDS, the planned
    # pipeline through HBM scratch (pc_large_kernel.hip) -- reference tiling of a 480 x 480 crop at other sample_point_size settings
    "p60": dict(kind="fft", h=480, w=480, n=60, grid=(8, 8), origin=(0, 0), stride=(60, 60), batch=1024, s=7,
                name="p60: FftMethod 480x480, 8x8 grid of 60x60 patches (half-tile kernel since r05; MOF_FFT_HALF=0: the full-tile planned kernel), batch=1024 per GPU",
                bytes_per_pair=2 * 480 * 480 + 64 * 8),
    "p96": dict(kind="fft", h=480, w=480, n=96, grid=(5, 5), origin=(0, 0), stride=(96, 96), batch=1024, s=12,
                name="p96: FftMethod 480x480, 5x5 grid of 96x96 patches (half-tile kernel since r05; MOF_FFT_HALF=0: the full-tile planned kernel), batch=1024 per GPU",
                bytes_per_pair=2 * 480 * 480 + 25 * 8),
    "p62": dict(kind="fft", h=496, w=496, n=62, grid=(8, 8), origin=(0, 0), stride=(62, 62), batch=1024, s=7,
                name="p62: FftMethod 496x496, 8x8 grid of 62x62 patches padded to 64 (planned kernel), batch=1024 per GPU",
                bytes_per_pair=2 * 496 * 496 + 64 * 8),
    # front ends of the planned kernel (r06: four pixels per load on every one of them): the long-range mode on ONE quarter-resolution patch of
    # 60 / 96 pixels (frames of 240 / 384 pixels) against the gray compile-time-plan kernel on one patch of the same size, and p60 on BGR8
    "lr60": dict(kind="fftlr", h=240, w=240, n=60, grid=(4, 4), origin=(0, 0), stride=(60, 60), batch=4096, s=7,
                 name="lr60: processImageLongRange on 240x240 frames (one quarter-resolution 60x60 patch), batch=4096", bytes_per_pair=2 * 240 * 120 + 8),
    "lr96": dict(kind="fftlr", h=384, w=384, n=96, grid=(4, 4), origin=(0, 0), stride=(96, 96), batch=4096, s=12,
                 name="lr96: processImageLongRange on 384x384 frames (one quarter-resolution 96x96 patch), batch=4096", bytes_per_pair=2 * 384 * 192 + 8),
    "g60": dict(kind="fft", h=60, w=60, n=60, grid=(1, 1), origin=(0, 0), stride=(60, 60), batch=4096, s=7,
                name="g60: FftMethod on 60x60 gray frames, ONE 60x60 patch (MOF_FFT_HALF=0: the planned kernel lr60 compares with), batch=4096", bytes_per_pair=2 * 60 * 60 + 8),
    "g96": dict(kind="fft", h=96, w=96, n=96, grid=(1, 1), origin=(0, 0), stride=(96, 96), batch=4096, s=12,
                name="g96: FftMethod on 96x96 gray frames, ONE 96x96 patch (MOF_FFT_HALF=0: the planned kernel lr96 compares with), batch=4096", bytes_per_pair=2 * 96 * 96 + 8),
    "p54": dict(kind="fft", h=486, w=486, n=54, grid=(9, 9), origin=(0, 0), stride=(54, 54), batch=1024, s=6,
                name="p54: FftMethod 486x486, 9x9 grid of 54x54 patches (full-tile planned kernel), batch=1024 per GPU", bytes_per_pair=2 * 486 * 486 + 81 * 8),
    "p54bgr": dict(kind="fft", bgr=True, h=486, w=486, n=54, grid=(9, 9), origin=(0, 0), stride=(54, 54), batch=512, s=6,
                   name="p54bgr: p54 on interleaved BGR8 frames (CV_RGB2GRAY fused into the planned kernel's four-pixel loads), batch=512 per GPU",
                   bytes_per_pair=3 * 2 * 486 * 486 + 81 * 8),
    "l160": dict(kind="fft", h=480, w=480, n=160, grid=(3, 3), origin=(0, 0), stride=(160, 160), batch=512, s=15,
                 name="l160: FftMethod 480x480, 3x3 grid of 160x160 patches (fused half-tile kernel; MOF_FFT_HALF=0: the pipeline through HBM scratch), batch=512 per GPU",
                 bytes_per_pair=2 * 480 * 480 + 9 * 8),
    # the next size cliff: padded sizes 193 .. 256 -- half a tile (M x (M/2 + 8) complex) no longer fits one CU's LDS
    "l200": dict(kind="fft", h=480, w=480, n=200, grid=(2, 2), origin=(0, 0), stride=(200, 200), batch=512, s=15,
                 name="l200: FftMethod 480x480, 2x2 grid of 200x200 patches (transform size 200), batch=512 per GPU",
                 bytes_per_pair=2 * 4 * 200 * 200 + 4 * 8),
    "l196": dict(kind="fft", h=480, w=480, n=196, grid=(2, 2), origin=(0, 0), stride=(200, 200), batch=512, s=15,
                 name="l196: FftMethod 480x480, 2x2 grid of 196x196 patches zero-padded to 200 (r06: the tuned transforms with the row kernel padding and the box-zero rule), batch=512 per GPU",
                 bytes_per_pair=2 * 4 * 196 * 196 + 4 * 8),
    "l320": dict(kind="fft", h=640, w=640, n=320, grid=(2, 2), origin=(0, 0), stride=(320, 320), batch=256, s=15,
                 name="l320: FftMethod 640x640, 2x2 grid of 320x320 patches (r06: tuned transforms 16 x 20), batch=256 per GPU",
                 bytes_per_pair=2 * 640 * 640 + 4 * 8),
    "l240": dict(kind="fft", h=480, w=480, n=240, grid=(2, 2), origin=(0, 0), stride=(240, 240), batch=512, s=15,
                 name="l240: FftMethod 480x480, 2x2 grid of 240x240 patches (transform size 240), batch=512 per GPU",
                 bytes_per_pair=2 * 480 * 480 + 4 * 8),
    # transform sizes whose two-stage plans end in an odd radix (10 x 25, 16 x 25, 16 x 27): tuned transforms with the real-only
    # slots of the spectrum taken from the images' exact integer sums (r06)
    "l250": dict(kind="fft", h=512, w=512, n=250, grid=(2, 2), origin=(0, 0), stride=(250, 250), batch=512, s=15,
                 name="l250: FftMethod 512x512, 2x2 grid of 250x250 patches (tuned transforms 10 x 25), batch=512 per GPU",
                 bytes_per_pair=2 * 4 * 250 * 250 + 4 * 8),
    "l400": dict(kind="fft", h=800, w=800, n=400, grid=(2, 2), origin=(0, 0), stride=(400, 400), batch=128, s=15,
                 name="l400: FftMethod 800x800, 2x2 grid of 400x400 patches (tuned transforms 16 x 25), batch=128 per GPU",
                 bytes_per_pair=2 * 800 * 800 + 4 * 8),
    "l432": dict(kind="fft", h=864, w=864, n=432, grid=(2, 2), origin=(0, 0), stride=(432, 432), batch=128, s=15,
                 name="l432: FftMethod 864x864, 2x2 grid of 432x432 patches (tuned transforms 16 x 27), batch=128 per GPU",
                 bytes_per_pair=2 * 864 * 864 + 4 * 8),
    # beyond 512: the tuned transforms with a first radix up to 32 (r06: 24 x 30)
    "l720": dict(kind="fft", h=720, w=720, n=720, grid=(1, 1), origin=(0, 0), stride=(720, 720), batch=128, s=15,
                 name="l720: FftMethod 720x720, ONE 720x720 patch (tuned transforms 24 x 30), batch=128 per GPU",
                 bytes_per_pair=2 * 720 * 720 + 8),
    "l480": dict(kind="fft", h=480, w=480, n=480, grid=(1, 1), origin=(0, 0), stride=(480, 480), batch=512, s=15,
                 name="l480: FftMethod 480x480, ONE 480x480 patch (the reference's whole-frame fallback), batch=512 per GPU",
                 bytes_per_pair=2 * 480 * 480 + 8),
    # (the t<n> workloads below are generated: one per remaining half-tile size, for the pitch sweeps of tools/sweep_half_pitch.sh)
    # the node's whole per-frame chain on the device: u8 frame pairs -> K1 shifts -> getRT (undistort,
    # RANSAC homography, decomposition, IMU-consistent pick) -> rotation + velocity; nothing but 64 B per pair leaves the GPU
    "refrt": dict(kind="fft+rt", h=480, w=480, n=120, grid=(4, 4), origin=(0, 0), stride=(120, 120), batch=1024, s=15,
                  name="refrt: ref + the getRT geometry tail on the device (frames -> rotation, velocity), batch=1024 per GPU",
                  bytes_per_pair=2 * 480 * 480 + 64),
    # the take-off mode: quarter-scale frames -> one long-range shift -> get2DT (closed form), on the device
    "reflr": dict(kind="fft+2dt", h=480, w=480, n=120, grid=(4, 4), origin=(0, 0), stride=(120, 120), batch=1024, s=15,
                  name="reflr: processImageLongRange (1/4-scale frames, one 120x120 patch) + get2DT on the device, batch=1024 per GPU",
                  # cv::resize(1/4, INTER_LINEAR) taps pixels (4x+1, 4x+2) x (4y+1, 4y+2): half of the rows are needed (whole
                  # 64-byte sectors of them), the other half never leaves HBM
                  bytes_per_pair=2 * 480 * 240 + 64),
    # c2 with the node's front end fused in: interleaved BGR8 frames, CV_RGB2GRAY inside the load
    "c2bgr": dict(kind="fft", bgr=True, h=480, w=752, n=64, grid=(8, 8), origin=(1, 1), stride=(98, 59), batch=512, s=8,
                  name="c2bgr: c2 on interleaved BGR8 frames, CV_RGB2GRAY fused into the load, batch=512 per GPU",
                  bytes_per_pair=3 * min(2 * 64 * 64 * 64, 2 * 752 * 480) + 64 * 8),
    # calibration of the FETCH_SIZE counter for this kernel's access pattern: the 64x64 patches tile the frame
    # exactly, every frame byte is read exactly once per launch -> known HBM read bytes = 2*512*512 per pair
    "cal": dict(kind="fft", h=512, w=512, n=64, grid=(8, 8), origin=(0, 0), stride=(64, 64), batch=1024, s=8,
                name="cal: FftMethod 512x512 tiled exactly by 8x8 patches of 64x64 (counter calibration), batch=1024",
                bytes_per_pair=2 * 512 * 512 + 64 * 8),
    # ... and for the long-range load path (DS = 4: two 64-byte runs per lane and tapped row pair): 512 x 512 frames, pitch 512,
    # ONE quarter-resolution 128 x 128 patch = the whole frame; cv::resize(1/4) taps rows 4r+1, 4r+2 only, whole 128-byte
    # lines of them -> known HBM read bytes = 2 images * 256 rows * 512 B per pair
    "callr": dict(kind="fft+2dt", h=512, w=512, n=128, grid=(4, 4), origin=(0, 0), stride=(128, 128), batch=1024, s=15,
                  name="callr: long-range mode on 512x512 frames, one 128x128 quarter-resolution patch (counter calibration of the DS=4 load), batch=1024",
                  bytes_per_pair=2 * 256 * 512 + 64),
    "c5": dict(kind="fft+sr", h=480, w=752, n=64, grid=(8, 8), origin=(1, 1), stride=(98, 59), batch=1024, s=8,
               sr_res=480, sr_m=49.9, sr_x0=136,
               name="c5: FftMethod (c2) + scaleRotationEstimator on the 480x480 centre crop (log-polar M=49.9 + whole-frame "
                    "phase correlation), 752x480, batch=1024 frame pairs per GPU",
               # 2*W*H u8 in + flow vectors + (scale, rot)
               bytes_per_pair=2 * 752 * 480 + 64 * 8 + 8),
    # c2 on a VIDEO (FftMethod.cpp:1872: every frame is cur once and prev once): B + 1 frames = B consecutive pairs through the
    # sequence kernel (one real 2-D transform per frame and patch, the previous spectrum in registers)
    "c2seq": dict(kind="fftseq", h=480, w=752, n=64, grid=(8, 8), origin=(1, 1), stride=(98, 59), batch=1024, s=8,
                  name="c2seq: c2 on a video -- FftMethod on 1024 consecutive frame pairs (1025 frames per GPU), sequence kernel",
                  # one NEW frame's patch pixels per pair + flow vectors
                  bytes_per_pair=64 * 64 * 64 + 64 * 8),
    # c4 on a video: 128 x 128 patches, the half-tile sequence kernel
    "c4seq": dict(kind="fftseq", h=1080, w=1920, n=128, grid=(16, 16), origin=(0, 0), stride=(119, 63), batch=512, s=16,
                  name="c4seq: c4 on a video -- FftMethod 1920x1080, 16x16 grid of 128x128 patches, 512 consecutive frame pairs",
                  bytes_per_pair=1920 * 1080 + 256 * 8),
    # the reference's default geometry on a video (r05): the half-tile kernel's sequence form, a frame's spectrum kept in registers
    "refseq": dict(kind="fftseq", h=480, w=480, n=120, grid=(4, 4), origin=(0, 0), stride=(120, 120), batch=1024, s=15,
                   name="refseq: ref on a video -- FftMethod 480x480, 4x4 grid of 120x120 patches, 1024 consecutive frame pairs (1025 frames)",
                   bytes_per_pair=480 * 480 + 16 * 8),
    "l200seq": dict(kind="fftseq", h=480, w=480, n=200, grid=(2, 2), origin=(0, 0), stride=(200, 200), batch=512, s=20,
                    name="l200seq: l200 on a video -- 2x2 grid of 200x200 patches, 512 consecutive frame pairs (r06: every frame's row spectra formed once)",
                    bytes_per_pair=4 * 200 * 200 + 4 * 8),
    "l480seq": dict(kind="fftseq", h=480, w=480, n=480, grid=(1, 1), origin=(0, 0), stride=(480, 480), batch=512, s=20,
                    name="l480seq: l480 on a video -- ONE 480x480 patch, 512 consecutive frame pairs",
                    bytes_per_pair=480 * 480 + 8),
    "l160seq": dict(kind="fftseq", h=480, w=480, n=160, grid=(3, 3), origin=(0, 0), stride=(160, 160), batch=512, s=20,
                    name="l160seq: l160 on a video -- 3x3 grid of 160x160 patches, 512 consecutive frame pairs",
                    bytes_per_pair=480 * 480 + 9 * 8),
    # c5 on a VIDEO (the node's real workload, scaleRotationEstimator.cpp:34-148 steady state): B + 1 consecutive frames,
    # K1 on the B consecutive pairs, the estimator in sequence mode (every frame remapped and row-transformed once)
    "c5seq": dict(kind="fft+srseq", h=480, w=752, n=64, grid=(8, 8), origin=(1, 1), stride=(98, 59), batch=1024, s=8,
                  sr_res=480, sr_m=49.9, sr_x0=136,
                  name="c5seq: c5 on a video -- FftMethod on 1024 consecutive frame pairs + scaleRotationEstimator in sequence "
                       "mode (first frame INTER_CUBIC, then INTER_LANCZOS4 once per frame), 752x480, 1025 frames per GPU",
                  # one NEW frame per pair (the other one was the previous pair's) + flow vectors + (scale, rot)
                  bytes_per_pair=752 * 480 + 64 * 8 + 8),
    # BASELINE c1 geometry (the CPU plumbing config) on the GPU: BlockMethod, 272x272 crop, 8x8 blocks of 32x32, +-8 px
    "c1": dict(kind="bm", block_method=True, h=272, w=272, block=32, step=0, radius=8, batch=1024, s=6,
               name="c1: BlockMethod 272x272 crop, 8x8 grid of 32x32 blocks, scanRadius=8, batch=1024 per GPU",
               bytes_per_pair=65536 + 73984 + 130),
    # the reference's own block-matching defaults (config/default.yaml:29-32): scan_radius 21, step_size 24,
    # sample_point_size 120 on the 752x480 camera frame -> 4 x 3 blocks of 120 x 120, 43 x 43 candidate shifts each
    "bmref": dict(kind="bm", h=480, w=752, block=120, step=24, radius=21, batch=256, s=12,
                  name="bmref: FastSpacedBMMethod 752x480, samplePointSize=120, stepSize=24, scanRadius=21 (reference default.yaml), batch=256 per GPU",
                  bytes_per_pair=12 * 120 * 120 + (4 * 144 - 24 + 42) * (3 * 144 - 24 + 42) + 2 * 12 + 2),
    # c3 with the node's front end fused in: interleaved BGR8 frames, CV_RGB2GRAY inside the staging loads
    "c3bgr": dict(kind="bm", bgr=True, h=480, w=752, block=16, step=8, radius=16, batch=512, s=12,
                  name="c3bgr: c3 on interleaved BGR8 frames, CV_RGB2GRAY fused into the staging loads, batch=512 per GPU",
                  bytes_per_pair=3 * (540 * 256 + 744 * 456) + 2 * 540 + 2),
    "c3": dict(kind="bm", h=480, w=752, block=16, step=8, radius=16, batch=1024, s=12,
               name="c3: FastSpacedBMMethod 752x480, samplePointSize=16, stepSize=8, scanRadius=16, batch=1024 per GPU",
               # blocks*sps^2 + window area + 2*blocks + 2
               bytes_per_pair=540 * 256 + 744 * 456 + 2 * 540 + 2),
}


def cpu_baseline(wl, budget_s: float = 12.0):
    """Time the CPU oracle (1 thread, like the reference's serial patch loop FftMethod.cpp:1829-1866)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from mrs_optic_flow_amd import synth

    n_gen = 4
    cur, prev, _, _ = synth.batch_np(n_gen, wl["h"], wl["w"], wl["s"], classes=False, k0=1)
    if wl["kind"] == "fft+srseq":
        lay = O.fft_layout(wl["w"], wl["h"], wl["n"], wl["grid"][0], wl["grid"][1], wl["origin"], wl["stride"])
        x0, r = wl["sr_x0"], wl["sr_res"]
        seq_est = O.ScaleRotationEstimator(r, wl["sr_m"], 32)
        seq_est.processImage(prev[0][:r, x0:x0 + r])

        def run(k):  # one new frame: K1 against the previous frame, one stateful estimator call
            O.fft_process(cur[k % n_gen], prev[k % n_gen], lay, 32)
            seq_est.processImage(cur[k % n_gen][:r, x0:x0 + r])
        what = "f32 oracle (oracle/pc_ref.c + lp_ref.c, stateful estimator: one remap + correlation per frame)"
    elif wl["kind"] == "fft+sr":
        lay = O.fft_layout(wl["w"], wl["h"], wl["n"], wl["grid"][0], wl["grid"][1], wl["origin"], wl["stride"])
        x0, r = wl["sr_x0"], wl["sr_res"]

        def run(k):
            O.fft_process(cur[k % n_gen], prev[k % n_gen], lay, 32)
            est = O.ScaleRotationEstimator(r, wl["sr_m"], 32)
            est.processImage(prev[k % n_gen][:r, x0:x0 + r])
            est.processImage(cur[k % n_gen][:r, x0:x0 + r])
        what = "f32 oracle (oracle/pc_ref.c + lp_ref.c)"
    elif wl["kind"] == "fft+2dt":
        lay = O.fft_layout(wl["w"], wl["h"], wl["n"], wl["grid"][0], wl["grid"][1], wl["origin"], wl["stride"])
        ocam = O.GeomCamera(400.0, 400.0, 240.0, 240.0, -0.01, 0.002, 0.0, 0.0, 0.0)
        ol = O.GeomLayout(1, 1, 0, 0, wl["n"], wl["n"], wl["n"])
        opar = O.Geom2dtParams(3.0, 0.02, 0.01, -0.02, 0.3)

        def run(k):
            flow, _ = O.fft_process_long_range(cur[k % n_gen], prev[k % n_gen], lay, 32)
            O.geom_get_2dt(flow, ol, ocam, opar)
        what = "f32 oracle long-range (oracle/pc_ref.c) + get2DT (oracle/geom_ref.c)"
    elif wl["kind"] == "fft+rt":
        import ctypes as C
        lay = O.fft_layout(wl["w"], wl["h"], wl["n"], wl["grid"][0], wl["grid"][1], wl["origin"], wl["stride"])
        ocam = O.GeomCamera(400.0, 400.0, 240.0, 240.0, -0.01, 0.002, 0.0, 0.0, 0.0)
        ol = O.GeomLayout(wl["grid"][0], wl["grid"][1], wl["origin"][0], wl["origin"][1], wl["stride"][0], wl["stride"][1], wl["n"])
        ident = (C.c_double * 4)(0, 0, 0, 1)
        opar = O.GeomRtParams(3.0, 0.02, 0.0, ident, ident, (C.c_double * 3)(0, 0, 0))

        def run(k):
            flow, _ = O.fft_process(cur[k % n_gen], prev[k % n_gen], lay, 32)
            O.geom_get_rt(flow, ol, ocam, opar, 8)
        what = "f32 oracle (oracle/pc_ref.c) + fp64 getRT (oracle/geom_ref.c)"
    elif wl["kind"] in ("fft", "fftseq"):  # (the CPU path transforms both patches of every pair, as cv::phaseCorrelate does)
        lay = O.fft_layout(wl["w"], wl["h"], wl["n"], wl["grid"][0], wl["grid"][1], wl["origin"], wl["stride"])
        run = lambda k: O.fft_process(cur[k % n_gen], prev[k % n_gen], lay, 32)
        what = "f32 oracle (oracle/pc_ref.c)"
    else:
        cfg = (O.bm_config_block_method(wl["h"], wl["block"], wl["radius"]) if wl.get("block_method")
               else O.bm_config_fast_spaced(wl["w"], wl["h"], wl["block"], wl["step"], wl["radius"]))
        run = lambda k: O.bm_process(cur[k % n_gen], prev[k % n_gen], cfg)
        what = "integer oracle (oracle/bm_ref.c)"
    run(0)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:
        run(done)
        done += 1
    dt = time.perf_counter() - t0
    res = {"value": done / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{done} frame pairs of the same workload in {dt:.1f} s, {what}, 1 thread, gcc -O2"}
    if wl["kind"] in ("fft", "fftseq") and wl["n"] in (32, 64, 128):
        res["tuned"] = cpu_baseline_tuned(wl, cur, prev, lay, O)
    # the same port over every host core of this box (ctypes releases the GIL), for scale only
    try:
        from concurrent.futures import ThreadPoolExecutor

        cores = min(len(os.sched_getaffinity(0)), 16)  # a 1-GPU box grants 16 host cores
        n_all = max(cores * 2, int(done / dt * cores * 4))
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(run, range(n_all)))
        res["all_cores"] = {"value": n_all / (time.perf_counter() - t1), "cores": cores}
    except Exception:
        pass
    return res


def cpu_baseline_tuned(wl, cur, prev, lay, O, budget_s: float = 6.0):
    """A FAIR CPU leg (review item): the same estimator written the way a fast CPU library would (oracle/pc_fast.c --
    iterative radix-4 Stockham over batches of lines, precomputed twiddles, real-input half spectra, no per-call
    allocation), rebuilt with -O3 -march=native on this box when gcc is there, checked against the f32 oracle on the very
    sample it times, one thread and all cores. The oracle above stays the checker; this is only timed."""
    import subprocess
    import tempfile

    import numpy as np

    flags, path = "gcc -O3 (portable prebuilt)", None
    try:
        tmp = tempfile.mkdtemp(prefix="pcfast_")
        out = os.path.join(tmp, "libpcfast_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu99", "-fPIC", "-shared", "-I", os.path.join(ROOT, "oracle"),
                               "-o", out, os.path.join(ROOT, "oracle", "pc_fast.c"), "-lm"], stderr=subprocess.DEVNULL)
        path, flags = out, "gcc -O3 -march=native (built on this box)"
    except Exception:
        pass
    O.fast_lib(path)
    n_gen = cur.shape[0]
    worst = 0.0
    for k in range(n_gen):  # agreement with the checker on the sample that is timed
        want, _ = O.fft_process(cur[k], prev[k], lay, 32)
        got = O.fft_process_fast(cur[k], prev[k], lay)
        if not np.array_equal(np.isnan(got), np.isnan(want)):
            return {"error": "tuned CPU path disagrees with the oracle on validity"}
        worst = max(worst, float(np.nanmax(np.abs(got - want))) if np.isfinite(want).any() else 0.0)
    if worst > 1e-4:
        return {"error": f"tuned CPU path differs from the f32 oracle by {worst:.3g} px"}
    run = lambda k: O.fft_process_fast(cur[k % n_gen], prev[k % n_gen], lay)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:
        run(done)
        done += 1
    dt = time.perf_counter() - t0
    res = {"value": done / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port-tuned",
           "sample": f"{done} frame pairs in {dt:.1f} s, oracle/pc_fast.c, {flags}, 1 thread",
           "max_abs_diff_vs_f32_oracle_px": worst}
    try:
        from concurrent.futures import ThreadPoolExecutor

        cores = min(len(os.sched_getaffinity(0)), 16)
        n_all = max(cores * 4, int(done / dt * cores * 3))
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(run, range(n_all)))
        res["all_cores"] = {"value": n_all / (time.perf_counter() - t1), "cores": cores}
    except Exception:
        pass
    return res


def load_traffic(tag: str):
    """HBM bytes per launch from a committed rocprofv3 --pmc run of this command (profiles/traffic_*.json)."""
    path = os.path.join(ROOT, "profiles", f"traffic_{tag}.json")
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def build_workload(wl, dev, local_rank: int, rank: int, graph: bool = False):
    """Generates the HBM-resident batch of one workload and returns (launch, engine, out_buffer_setter).
    `launch()` runs one step (one pass of the hot path over the batch) on torch's current stream."""
    import torch

    from mrs_optic_flow_amd import FastSpacedBMMethod, FftMethod, ScaleRotationEstimator, synth

    B = wl["batch"]
    state = {"out": None}
    if wl["kind"] in ("fft+srseq", "fftseq"):
        # every rank owns its own video (texture index = rank): B + 1 frames = B consecutive pairs
        video, _ = synth.video_torch(B + 1, wl["h"], wl["w"], dev, k=rank)
        cur, prev = video[1:], video[:-1]  # a video needs no copy: cur = frames + 1, prev = frames
    else:
        # every rank owns its own shard of the global batch: pairs [rank*B, (rank+1)*B)
        cur, prev, _, _ = synth.batch_torch(B, wl["h"], wl["w"], wl["s"], dev, k0=rank * B)
    if wl["kind"] in ("fft", "fftlr", "fftseq", "fft+sr", "fft+srseq", "fft+rt", "fft+2dt"):
        eng = FftMethod(sample_point_size=wl["n"], frame_shape=(wl["h"], wl["w"]), grid=wl["grid"],
                        origin=wl["origin"], stride=wl["stride"], device=local_rank)
        state["out"] = torch.empty((B, eng.n_patches, 2), dtype=torch.float64, device=dev)
        if wl["kind"] == "fftseq":
            def launch():
                eng.process_sequence_device(video, out=state["out"])
                return state["out"]
        elif wl["kind"] == "fft+srseq":
            sr = ScaleRotationEstimator(wl["sr_res"], wl["sr_m"], device=local_rank, batch_chunk=1024)  # 1024-pair passes: +1.7 % for 3.8 GB of scratch (the library default is 512)
            x0, r = wl["sr_x0"], wl["sr_res"]
            crop = video[:, :r, x0:x0 + r]
            sr.process_sequence_device(crop[:2])  # arm the steady state: every timed frame goes through INTER_LANCZOS4

            side = torch.cuda.Stream(device=dev) if os.environ.get("MOF_BENCH_FFT_SIDE_STREAM", "0") != "0" else None

            def launch():
                cs = torch.cuda.current_stream(dev)
                if side is not None:
                    side.wait_stream(cs)
                eng.process_sequence_device(video, out=state["out"], stream=side)
                srout = sr.process_sequence_device(crop[1:], resolve_gate=False)  # B new frames = B pairs
                if side is not None:
                    cs.wait_stream(side)
                return torch.cat([state["out"].reshape(B, -1), srout], dim=1)
        elif wl["kind"] == "fft+sr":
            sr = ScaleRotationEstimator(wl["sr_res"], wl["sr_m"], device=local_rank, batch_chunk=1024)  # 1024-pair passes: +1.7 % for 3.8 GB of scratch (the library default is 512)
            x0, r = wl["sr_x0"], wl["sr_res"]
            cur_c, prev_c = cur[:, :r, x0:x0 + r], prev[:, :r, x0:x0 + r]

            side = torch.cuda.Stream(device=dev) if os.environ.get("MOF_BENCH_FFT_SIDE_STREAM", "0") != "0" else None

            def launch():
                cs = torch.cuda.current_stream(dev)
                if side is not None:
                    side.wait_stream(cs)
                eng.process_batch_device(cur, prev, out=state["out"], stream=side)
                srout = sr.process_batch_device(cur_c, prev_c)
                if side is not None:
                    cs.wait_stream(side)
                return torch.cat([state["out"].reshape(B, -1), srout], dim=1)
        elif wl["kind"] == "fftlr":
            def launch():
                return eng.process_long_range_batch_device(cur, prev)
        elif wl["kind"] == "fft+2dt":
            import numpy as np

            from mrs_optic_flow_amd import geometry as G
            gcam = G.Camera(400.0, 400.0, 240.0, 240.0, -0.01, 0.002, 0.0, 0.0, 0.0)
            n_lr = eng._lib.mof_fft_long_range_patches(eng._h)
            gl = G.Layout(1, 1, 0, 0, wl["n"], wl["n"], wl["n"]) if n_lr == 1 else None  # the quarter image is one patch
            assert gl is not None, "reflr expects the reference geometry (one long-range patch)"
            row = np.frombuffer(bytes(G.T2dParams(3.0, 0.02, 0.01, -0.02, 0.3)), dtype=np.float64).copy()
            d_par = torch.from_numpy(np.repeat(row[None, :], B, axis=0)).to(dev)

            def launch():
                flow = eng.process_long_range_batch_device(cur, prev)
                return G.get_2dt_batch_device(flow, gl, gcam, d_par)
        elif wl["kind"] == "fft+rt":
            import ctypes as C

            import numpy as np

            from mrs_optic_flow_amd import geometry as G
            gcam = G.Camera(400.0, 400.0, 240.0, 240.0, -0.01, 0.002, 0.0, 0.0, 0.0)
            gl = G.Layout(wl["grid"][0], wl["grid"][1], wl["origin"][0], wl["origin"][1], wl["stride"][0], wl["stride"][1], wl["n"])
            ident = (C.c_double * 4)(0, 0, 0, 1)
            par = G.RtParams(3.0, 0.02, 0.0, ident, ident, (C.c_double * 3)(0, 0, 0))
            row = np.frombuffer(bytes(par), dtype=np.float64).copy()
            d_par = torch.from_numpy(np.repeat(row[None, :], B, axis=0)).to(dev)

            def launch():  # (running the tail on a second stream under the next batch's K1 gains nothing: measured r02,
                # its single-lane fp64 stretches still take whole VALU issue slots and K1 slows by the tail's duration)
                eng.process_batch_device(cur, prev, out=state["out"])
                return G.get_rt_batch_device(state["out"], gl, gcam, d_par, 8)
        elif wl.get("bgr"):
            # synthetic colour frames: three different affine maps of the gray texture (data stays u8)
            def colour(g):
                g16 = g.to(torch.int16)
                return torch.stack([g, (255 - g16 // 2).to(torch.uint8), (g16 * 3 // 4 + 20).to(torch.uint8)], dim=-1).contiguous()
            cur3, prev3 = colour(cur), colour(prev)

            def launch():
                return eng.process_batch_device_bgr(cur3, prev3)
        else:
            def launch():
                eng.process_batch_device(cur, prev, out=state["out"])
                return state["out"]
    else:
        if wl.get("block_method"):
            from mrs_optic_flow_amd import BlockMethod
            eng = BlockMethod(wl["h"], wl["block"], wl["radius"], device=local_rank)
        else:
            eng = FastSpacedBMMethod(wl["block"], wl["radius"], wl["step"], (wl["h"], wl["w"]), device=local_rank)

        if wl.get("bgr"):
            def colour(g):
                g16 = g.to(torch.int16)
                return torch.stack([g, (255 - g16 // 2).to(torch.uint8), (g16 * 3 // 4 + 20).to(torch.uint8)], dim=-1).contiguous()
            cur3, prev3 = colour(cur), colour(prev)

            def launch():
                return eng.process_batch_device_bgr(cur3, prev3)[2]
        else:
            def launch():
                return eng.process_batch_device(cur, prev)[2]

    if graph:
        eager_launch = launch
        eager_launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                graph_result = eager_launch()

        def launch():  # (the captured engines are pinned by the library and by engine._CAPTURED until release_captured())
            g.replay()
            return graph_result

    return launch, eng, state


def timed_steps(step, steps: int, world: int, dev):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks. Returns seconds."""
    import torch
    import torch.distributed as dist

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    step(None, drain=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def roofline_block(tag: str, wl, B: int, kern_ms: float):
    bytes_per_launch = wl["bytes_per_pair"] * B
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    traffic = load_traffic(tag)
    blk = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
           # `traffic` is NOT measured by this run: PMC counters need their own rocprofv3 pass (tools/pmc.sh); the
           # number is the committed result of that pass for this command (per launch of `batch` pairs)
           "traffic_source": (f"profiles/traffic_{tag}.json: fabric bytes of ALL the library's kernels in one step, from "
                              "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile_all.sh, "
                              "tools/summarize_round.py; FETCH_SIZE doubled for the kernels a known byte count shows it halves); "
                              "not measured live") if traffic is not None else None,
           "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
           "binding": binding_note(tag, wl)}
    if wl["kind"] in ("fft", "fftseq"):
        # the resource that actually binds K1: vector fp32. Informational, next to the HBM figure.
        fl = fft_flops_per_pair(wl["n"], wl["grid"][0] * wl["grid"][1]) * B
        blk["compute"] = {"unit": "TFLOP/s", "achieved": fl / (kern_ms * 1e-3) / 1e12, "peak": VALU_FP32_PEAK_TFLOPS,
                          "frac": fl / (kern_ms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS, "flop_per_launch": fl,
                          "pipe": COMPUTE_PIPE_NOTE}
    return blk


OTHERS_SETTLE_S = 0.4


def measure_other(tag: str, dev, steps: int, warmup: int):
    """Compact record of one more BASELINE workload (N=1): value, kernel time, HBM fraction."""
    import torch

    wl = dict(WORKLOADS[tag])
    launch, eng, _ = build_workload(wl, dev, dev.index or 0, 0)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    # (settle: OTHERS_SETTLE_S of untimed launches first -- each of these records starts on a GPU that idled through the previous one's
    # set-up, and 20 - 50 steps of a 1 ms kernel would read its clock ramp, as the headline's burst did until r04)
    torch.cuda.synchronize()
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < OTHERS_SETTLE_S:
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
    for _ in range(warmup):
        launch()

    def step(i, drain=False):
        if drain:
            return
        ev0[i].record()
        launch()
        ev1[i].record()

    elapsed = timed_steps(step, steps, 1, dev)
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / steps
    B = wl["batch"]
    rec = {"workload": wl["name"], "value": B * steps / elapsed, "unit": "frame-pairs/s", "steps": steps,
           "warmup": warmup, "settle_s": OTHERS_SETTLE_S, "ms_per_step": elapsed / steps * 1e3, "kernel_ms": kern_ms,
           "frac": wl["bytes_per_pair"] * B / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "algorithmic_bytes_per_launch": wl["bytes_per_pair"] * B, "traffic": load_traffic(tag)}
    del launch, eng
    torch.cuda.empty_cache()
    return rec


for _n in (100, 144, 150, 162, 180, 192):
    WORKLOADS[f"t{_n}"] = _tile_workload(_n)


def host_entries_record(n=512):
    """The PCIe-INCLUSIVE rate of the host-pointer batch entry at c2 (mof_fft_process_batch_host, csrc/host_pipe.hpp: a three-slot upload /
    run / download pipeline over the device entry; host frames in, host results out) -- never `value`, which starts with the batch in HBM.
    Four memory layouts: pageable / pinned frames, as two pair buffers or as one video (cur = prev + one frame: each frame uploaded once)."""
    import numpy as np

    from mrs_optic_flow_amd import FftMethod, pinned_empty, synth
    w = WORKLOADS["c2"]
    fm = FftMethod(sample_point_size=w["n"], frame_shape=(w["h"], w["w"]), grid=w["grid"], origin=w["origin"], stride=w["stride"])
    base, _, _, _ = synth.batch_np(64, w["h"], w["w"], w["s"], classes=False, k0=3)
    frames = np.ascontiguousarray(np.tile(base, ((n + 64) // 64, 1, 1))[: n + 1])
    pin = pinned_empty(frames.shape)
    pin[:] = frames
    pc, pp = pinned_empty((n,) + frames.shape[1:]), pinned_empty((n,) + frames.shape[1:])
    pc[:] = frames[1:]
    pp[:] = frames[:-1]
    rec = {"workload": "c2", "pairs_per_call": n, "unit": "frame pairs/s, host frames in -> host results out (PCIe-inclusive; best of 3 calls)"}
    for label, c, p, up in (("pageable_pairs", frames[1:].copy(), frames[:-1].copy(), 2 * n), ("pageable_video", frames[1:], frames[:-1], n + 1),
                            ("pinned_pairs", pc, pp, 2 * n), ("pinned_video", pin[1:], pin[:-1], n + 1)):
        fm.process_batch_host(c, p)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            fm.process_batch_host(c, p)
            best = min(best, time.perf_counter() - t0)
        rec[label] = {"value": round(n / best, 1), "frames_gb_per_s": round(up * w["h"] * w["w"] / best / 1e9, 2)}
    return rec


def native_group_run(tag: str, wl, n_dev: int, steps: int, warmup: int, share_gpu: bool, sync_per_step: bool = False, settle_s: float = 0.4):
    """The batched-frames mode through the NATIVE shard group (mof_shard_fft_* / mof_shard_bm_*, csrc/mof_shard.hip): ONE process, no
    torch.distributed -- one engine and one HIP stream per device inside the library, ceil(B / G) contiguous shards, one in-place RCCL
    all-gather per step (ncclCommInitAll by mof_shard_*_init_gather). torch only allocates the per-device buffers and generates the
    shards. Weak scaling: `batch` pairs per device. --share-gpu: all shards on device 0 under the rehearsal knob
    MOF_SHARD_SHARE_DEVICE=1 (a one-GPU box; such a group cannot gather). Returns the record for the JSON line."""
    import ctypes as C

    import torch

    from mrs_optic_flow_amd import BlockMethod, FastSpacedBMMethod, FftMethod, _capi, synth

    if wl["kind"] not in ("fft", "bm") or wl.get("bgr"):
        raise SystemExit(f"--native serves the gray FftMethod / block-matching workloads (c2, c4, ref, c3, c1, ...), not {tag}")
    if share_gpu:
        os.environ["MOF_SHARD_SHARE_DEVICE"] = "1"
    devices = [0] * n_dev if share_gpu else list(range(n_dev))
    if not share_gpu and torch.cuda.device_count() < n_dev:
        raise SystemExit(f"--native --gpus {n_dev}: this process sees {torch.cuda.device_count()} device(s) (use --share-gpu to rehearse on one)")
    lib = _capi.load()
    B, G, fft = wl["batch"], n_dev, wl["kind"] == "fft"
    if fft:
        proto = FftMethod(sample_point_size=wl["n"], frame_shape=(wl["h"], wl["w"]), grid=wl["grid"], origin=wl["origin"], stride=wl["stride"], device=0)
        per_pair = proto.n_patches * 2
    elif wl.get("block_method"):
        proto = BlockMethod(wl["h"], wl["block"], wl["radius"], device=0)
    else:
        proto = FastSpacedBMMethod(wl["block"], wl["radius"], wl["step"], (wl["h"], wl["w"]), device=0)
    grp = C.c_void_p()
    dev_arr = (C.c_int * G)(*devices)
    create, destroy = (lib.mof_shard_fft_create, lib.mof_shard_fft_destroy) if fft else (lib.mof_shard_bm_create, lib.mof_shard_bm_destroy)
    process, sync = (lib.mof_shard_fft_process_batch_device, lib.mof_shard_fft_sync) if fft else (lib.mof_shard_bm_process_batch_device, lib.mof_shard_bm_sync)
    init_gather, ranks_of, stream_of = ((lib.mof_shard_fft_init_gather, lib.mof_shard_fft_gather_ranks, lib.mof_shard_fft_stream) if fft else
                                        (lib.mof_shard_bm_init_gather, lib.mof_shard_bm_gather_ranks, lib.mof_shard_bm_stream))
    _capi.check(create(C.byref(proto.cfg), dev_arr, G, C.byref(grp)))
    try:
        gather = 0 if share_gpu else 1
        if gather:
            _capi.check(init_gather(grp))
        rccl_ranks = int(ranks_of(grp))
        cur, prev, out, ext = [], [], [], []
        for i, d in enumerate(devices):
            dv = torch.device("cuda", d)
            with torch.cuda.device(dv):
                c, p, _, _ = synth.batch_torch(B, wl["h"], wl["w"], wl["s"], dv, k0=i * B)  # shard i of the global batch, generated on its own GPU
                cur.append(c)
                prev.append(p)
                if fft:
                    out.append(torch.empty((G * B, per_pair), dtype=torch.float64, device=dv))
                else:
                    out.append(torch.empty((G * int(lib.mof_shard_bm_slab_bytes(grp, G * B)),), dtype=torch.int8, device=dv))
                ext.append(torch.cuda.ExternalStream(int(stream_of(grp, i)), device=dv))
        pc, pp, po = ((C.c_void_p * G)(*[t.data_ptr() for t in ts]) for ts in (cur, prev, out))
        cs, ps, pitch = cur[0].stride(0), prev[0].stride(0), cur[0].stride(1)

        def sync_all():
            _capi.check(sync(grp))
            for d in set(devices):
                torch.cuda.synchronize(d)

        def step():
            _capi.check(process(grp, pc, cs, pp, ps, pitch, G * B, po, gather))
            if sync_per_step:
                _capi.check(sync(grp))

        sync_all()
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < settle_s:  # settled clocks, as every other record of this file
            for _ in range(10):
                step()
            sync_all()
        for _ in range(warmup):
            step()
        ev0 = [[torch.cuda.Event(enable_timing=True) for _ in devices] for _ in range(steps)]
        ev1 = [[torch.cuda.Event(enable_timing=True) for _ in devices] for _ in range(steps)]
        sync_all()
        t0 = time.perf_counter()
        for k in range(steps):
            for i in range(G):
                ev0[k][i].record(ext[i])
            step()
            for i in range(G):
                ev1[k][i].record(ext[i])
        sync_all()
        elapsed = time.perf_counter() - t0
        per_dev_ms = [sum(ev0[k][i].elapsed_time(ev1[k][i]) for k in range(steps)) / steps for i in range(G)]
        # the result of the whole batch, as device 0 holds it after the gather, against the plain engine on each shard (bit for bit)
        verified = True
        for i, d in enumerate(devices if gather else devices[:1]):
            with torch.cuda.device(d):
                if fft:
                    e = proto if d == 0 else FftMethod(sample_point_size=wl["n"], frame_shape=(wl["h"], wl["w"]), grid=wl["grid"], origin=wl["origin"], stride=wl["stride"], device=d)
                    want = e.process_batch_device(cur[i], prev[i]).reshape(B, per_pair)
                    got = (out[0] if gather and d == 0 else out[i])[i * B:(i + 1) * B]
                    verified = verified and bool(torch.equal(got.to(want.device), want))
                else:
                    e = proto if d == 0 else (BlockMethod(wl["h"], wl["block"], wl["radius"], device=d) if wl.get("block_method")
                                              else FastSpacedBMMethod(wl["block"], wl["radius"], wl["step"], (wl["h"], wl["w"]), device=d))
                    dx, dy, mode = e.process_batch_device(cur[i], prev[i])
                    blocks, slab = dx[0].numel(), int(lib.mof_shard_bm_slab_bytes(grp, G * B))
                    got = (out[0] if gather and d == 0 else out[i])[i * slab:(i + 1) * slab].to(dx.device)  # dx | dy | mode planes of rank i
                    verified = (verified and bool(torch.equal(got[:B * blocks], dx.reshape(-1))) and bool(torch.equal(got[B * blocks:2 * B * blocks], dy.reshape(-1)))
                                and bool(torch.equal(got[2 * B * blocks:2 * B * blocks + 8 * B], mode.reshape(-1))))
        rec = {"value": G * B * steps / elapsed, "unit": "frame-pairs/s", "n_devices": G, "devices": devices, "steps": steps, "warmup": warmup,
               "ms_per_step": elapsed / steps * 1e3, "per_device_step_ms": per_dev_ms,
               "per_device_step_ms_note": "HIP events on each shard's own stream around one group call: the shard's kernel(s) + its leg of the all-gather",
               "rccl_ranks": rccl_ranks, "gather": ("in-place ncclAllGather per step inside one ncclGroupStart/End, %d rank(s) formed by ncclCommInitAll" % rccl_ranks) if gather
               else "none (shards share a device: RCCL needs one rank per device)",
               "sync": "per step" if sync_per_step else "end of the timed region (steps are enqueued back to back on the group's streams, as the plain path's)",
               "batch_per_device": B, "shard_results_equal_plain_engine": verified, "share_gpu": share_gpu}
        return rec
    finally:
        destroy(grp)


def self_launch(n_ranks: int) -> int:
    """One process per GPU, started by bench.py itself: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT are set for
    every child BEFORE anything in it touches a GPU (the children are fresh interpreters of this script with the same
    arguments; under torch.distributed.run those variables are already there and this function is never reached). Rank 0's
    stdout -- the JSON line -- is relayed; the other ranks' stdout goes to stderr. Returns non-zero if any rank failed (the
    remaining ranks are then ended by their exact PIDs, never by pattern)."""
    import socket
    import subprocess
    import time

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
                for q in alive:  # a rank died: the others would wait for it in the next collective
                    q.terminate()
        time.sleep(0.05)
    return rc


_JSON_FD = None


def quiet_stdout() -> None:
    """From here on file descriptor 1 is stderr for everybody (RCCL prints a five-line version banner to stdout when a communicator is
    built -- ncclCommInitAll of the native group, the first collective of torch.distributed); the ONE JSON line of the contract goes to the
    saved descriptor (emit)."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line) -> None:
    data = (json.dumps(line) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main_native(args) -> None:
    """`bench.py --native --gpus N`: the contract's JSON line, measured through the native shard group (no torch.distributed)."""
    quiet_stdout()
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    rec = native_group_run(args.workload, wl, args.gpus, args.steps, args.warmup, args.share_gpu, args.native_sync_per_step)
    kern_ms = max(rec["per_device_step_ms"])
    line = {"metric": "frame_pairs_per_s" + ("_fft_phase_corr" if wl["kind"] == "fft" else ("_block_method" if wl.get("block_method") else "_fast_spaced_bm")),
            "value": rec["value"], "unit": "frame-pairs/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if wl["kind"] == "bm" else "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "batch_per_gpu": wl["batch"], "frame": f'{wl["w"]}x{wl["h"]} u8',
                       "parallelism": f"native shard group x{args.gpus}" + (" (all shards on GPU 0: rehearsal)" if args.share_gpu else ""),
                       "gather": rec["gather"], "hip_graph": False},
            "rccl_ranks": rec["rccl_ranks"], "native_shard_group": rec}
    if not args.share_gpu:  # (G shards on one device time G batches on one GPU: no roofline of ONE launch to state)
        line["roofline"] = roofline_block(args.workload, wl, wl["batch"], kern_ms)
        line["roofline"]["kernel_ms_note"] = "slowest device's step by HIP events on its shard stream (kernel + its leg of the gather)"
    emit(line)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (default 200: a 20-step burst ends before the GPU's clocks have settled and reads 13 %% low)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frame pairs per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the compact records of the other BASELINE workloads (c3, c4, c5, ref) after the headline")
    ap.add_argument("--sustain-s", type=float, default=2.0,
                    help="after the timed steps, run back-to-back steps for at least this long and report that rate too")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl == RCCL; gloo only to rehearse several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--blocking-gather", action="store_true",
                    help="--gpus > 1: blocking all_gather_into_tensor per step instead of the double-buffered non-blocking one")
    ap.add_argument("--native", action="store_true",
                    help="time the NATIVE shard group (mof_shard_*: one process, one engine + stream per device, RCCL all-gather inside the library) "
                         "over --gpus devices instead of one torch.distributed rank per GPU; with --share-gpu all shards sit on GPU 0 (no gather)")
    ap.add_argument("--native-sync-per-step", action="store_true", help="--native: mof_shard_*_sync after every step instead of at the end")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph and replay it (helps the multi-launch c5 pipeline)")
    args = ap.parse_args()

    if args.native:
        return main_native(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run bare: this process becomes the launcher. It touches no GPU (nothing below has
        # imported torch or loaded the library yet), starts N fresh rank processes of this same script and relays rank 0's line.
        raise SystemExit(self_launch(args.gpus))

    quiet_stdout()
    import torch
    import torch.distributed as dist

    from mrs_optic_flow_amd import sharding

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    B = wl["batch"]
    launch, eng, state = build_workload(wl, dev, local_rank, rank, graph=bool(args.graph and world == 1))

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    # The one collective of the batched-frames mode: all-gather of the flow vectors. With RCCL it is issued
    # non-blocking and double-buffered, so it overlaps the next batch's kernels (it is ~1 MB per rank). Which gather
    # runs is decided HERE from the arguments, identically on every rank, and reported in the JSON line; a failure
    # of the chosen path ends the run (no silent per-rank fallback: ranks that disagree would dead-lock).
    ag = None
    gather = "none (1 rank)"
    if world > 1:
        if args.backend == "nccl" and wl["kind"] == "fft" and not wl.get("bgr") and not args.blocking_gather:
            ag = sharding.AsyncGather((B, eng.n_patches, 2), torch.float64, dev, B * world)
            gather = "non-blocking double-buffered all_gather_into_tensor (RCCL)"
        else:
            gather = ("blocking all_gather_into_tensor (RCCL)" if args.backend == "nccl"
                      else "blocking all_gather (gloo rehearsal, host staging)")

    def step(i=None, drain=False):
        if drain:
            if ag is not None:
                ag.drain()
            return None
        if ag is not None:
            state["out"] = ag.slot()
        if i is not None and i < len(ev0):
            ev0[i].record()
        res = launch()
        if i is not None and i < len(ev1):
            ev1[i].record()
        if ag is not None:
            ag.submit().done()  # nobody reads the gathered vectors here: released at once (sharding.Gathered)
            return None
        if world > 1:
            res = sharding.gather_results(res, B * world)
        return res

    # Order (r05): the SUSTAINED leg runs first -- >= sustain_s seconds of the same step back to back, reported under "sustained" --, then the
    # W warm-up steps and the K timed steps of the contract. The K-step burst lasts ~13 ms at c2 with the driver's K = 20: measured
    # first, as until r04, it read the clock ramp of a GPU that had just idled (BENCH_r04: 1.54 M in the burst, 1.73 M sustained in the
    # same run); after the sustained leg it reads the settled clocks a streaming job runs at. --sustain-s 0 (the A/B scripts) skips the leg.
    # (r06) BOTH regimes are reported: `cold_burst` = exactly the W warm-up + K timed steps the command line asks for, started on a GPU
    # that has idled for a second (what a reader of `--steps 20 --warmup 5` would expect; it reads the clock ramp), then the sustained leg,
    # then the contract's W + K steps on settled clocks (`value`). No headline moves by protocol alone: both numbers are in every line.
    cold = None
    if args.sustain_s > 0:
        torch.cuda.synchronize()
        time.sleep(1.0)
        for _ in range(args.warmup):
            step()
        step(None, drain=True)
        cold_elapsed = timed_steps(step, args.steps, world, dev)
        cold = {"value": B * world * args.steps / cold_elapsed, "ms_per_step": cold_elapsed / args.steps * 1e3,
                "kernel_ms": sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps, "steps": args.steps, "warmup": args.warmup,
                "order": "first: after 1 s of idling, before the sustained leg (the r01-r04 protocol)"}
    sustained = None
    if args.sustain_s > 0:
        for _ in range(3):
            step()
        step(None, drain=True)
        probe = timed_steps(lambda i, drain=False: step(None, drain=drain), 5, world, dev) / 5
        n_sus = max(args.steps, int(args.sustain_s / probe) + 1)
        if world > 1:  # every rank must run the same number of steps (collectives inside)
            t = torch.tensor([n_sus], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_sus = int(t.item())
        sus_elapsed = timed_steps(lambda i, drain=False: step(None, drain=drain), n_sus, world, dev)
        sustained = {"steps": n_sus, "seconds": sus_elapsed, "value": B * world * n_sus / sus_elapsed,
                     "ms_per_step": sus_elapsed / n_sus * 1e3, "order": "before the warm-up and the timed steps"}
    for _ in range(args.warmup):
        step()
    step(None, drain=True)
    elapsed = timed_steps(step, args.steps, world, dev)
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps  # HIP events on the launch stream

    if rank == 0:
        pairs = B * world * args.steps
        line = {
            "metric": "frame_pairs_per_s" + {"fft": "_fft_phase_corr", "fftlr": "_long_range", "fftseq": "_fft_phase_corr_sequence",
                                             "fft+sr": "_fft_phase_corr_plus_scale_rotation",
                                             "fft+srseq": "_fft_phase_corr_plus_scale_rotation_sequence",
                                             "fft+rt": "_fft_phase_corr_plus_get_rt",
                                             "fft+2dt": "_long_range_plus_get_2dt",
                                             "bm": "_block_method" if wl.get("block_method") else "_fast_spaced_bm"}[wl["kind"]],
            "value": pairs / elapsed,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if wl["kind"] == "bm" else "f32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "batch_per_gpu": B, "frame": f'{wl["w"]}x{wl["h"]} u8',
                       "parallelism": f"frame-pair shards x{world}, all-gather of flow vectors" if world > 1 else "1 GPU",
                       "gather": gather, "hip_graph": bool(args.graph and world == 1)},
            "roofline": roofline_block(args.workload, wl, B, kern_ms),
        }
        if cold is not None:
            line["cold_burst"] = cold
            line["value_regime"] = "settled clocks: the W + K steps run after the sustained leg (cold_burst = the same W + K steps from an idle GPU)"
        if sustained is not None:
            line["sustained"] = sustained
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl)
        if world == 1 and not args.no_others and args.workload == "c2" and not args.batch:
            del launch, eng, state, step
            torch.cuda.empty_cache()
            # driver-visible records of the other BASELINE configurations (same protocol, fewer steps)
            # the same workload through the NATIVE shard group (C++ host, RCCL inside the library) on this one device: a 1-rank gather
            try:
                line["native_shard_group"] = native_group_run("c2", dict(WORKLOADS["c2"]), 1, 100, 10, False)
            except BaseException as e:  # (never at the price of the headline line)
                line["native_shard_group"] = {"error": f"{type(e).__name__}: {e}"}
            line["other_workloads"] = {tag: measure_other(tag, dev, st, 5)
                                       for tag, st in (("c3", 50), ("c4", 20), ("c5", 40), ("c2seq", 50), ("c4seq", 10), ("c5seq", 40), ("ref", 50), ("refseq", 50),
                                                       ("bmref", 50), ("refrt", 50), ("reflr", 50),
                                                       ("p60", 50), ("l160", 40), ("l200", 20), ("l240", 20), ("l250", 20), ("l400", 20), ("l480", 20), ("l720", 20))}  # the planned kernel, the half-tile kernel (r05), the large-patch pipeline
            try:
                line["host_entries"] = host_entries_record()
            except BaseException as e:
                line["host_entries"] = {"error": f"{type(e).__name__}: {e}"}
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
