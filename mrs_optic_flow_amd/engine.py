"""Python handles on the C ABI engines, named after the reference's processors.

The reference's host language is C++ (the C++ mirror is include/mof/processors.hpp);
these classes exist so tests and bench.py read like calls on the reference's own
classes: ``FftMethod(frameSize, samplePointSize, max_px_speed).processImage(im)``
(/root/reference/include/FftMethod.h:434-439), ``BlockMethod``
(/root/reference/include/BlockMethod.h:40-42) and ``FastSpacedBMMethod``
(/root/reference/include/FastSpacedBMMethod_OCL.h:38-42). numpy uint8 arrays stand
in for ``cv::Mat``; torch device tensors are accepted by the batched calls, which hand
raw device pointers and the current HIP stream to the library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import BmConfig, FftConfig, SrConfig, check


def _np_u8(frame) -> np.ndarray:
    a = np.asarray(frame)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise ValueError("frame must be a 2-D uint8 array (CV_8UC1)")
    if a.strides[1] != 1:
        a = np.ascontiguousarray(a)
    return a


def _host_batch_views(cur, prev):
    """The two [n, H, W] uint8 arrays of a host batch as the C ABI takes them -- (cur, prev, cur_stride, prev_stride, pitch) -- WITHOUT a copy
    when their rows are dense runs of bytes with one pitch: a video handed over as ``frames[1:], frames[:-1]`` keeps its memory, which is
    how the library sees that it is one (include/mof.h, mof_fft_process_batch_host)."""
    cur, prev = np.asarray(cur), np.asarray(prev)
    if cur.shape != prev.shape or cur.ndim != 3:
        raise ValueError("cur and prev must be [n, H, W] arrays of one shape")

    def usable(a):
        return a.dtype == np.uint8 and a.strides[2] == 1 and a.strides[1] >= a.shape[2] and (a.shape[0] < 2 or a.strides[0] > 0)

    if not (usable(cur) and usable(prev) and cur.strides[1] == prev.strides[1]):
        cur, prev = np.ascontiguousarray(cur, dtype=np.uint8), np.ascontiguousarray(prev, dtype=np.uint8)
    return cur, prev, max(cur.strides[0], 0), max(prev.strides[0], 0), cur.strides[1]


def pinned_empty(shape, dtype=np.uint8) -> np.ndarray:
    """A numpy array in page-locked host memory (mof_host_alloc): frames kept in one are DMA'd by the *_batch_host entries from where they
    lie. The memory is returned when the array (and every view of it) is gone."""
    import weakref

    lib = _capi.load()
    dt = np.dtype(dtype)
    nbytes = max(1, int(np.prod(shape)) * dt.itemsize)
    p = C.c_void_p()
    check(lib.mof_host_alloc(nbytes, C.byref(p)))
    buf = (C.c_uint8 * nbytes).from_address(p.value)
    weakref.finalize(buf, lib.mof_host_free, C.c_void_p(p.value))
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))


# Engines that a HIP graph captured (include/mof.h, "HIP graphs"): a captured kernel node holds raw pointers into the
# engine's device memory, so the Python handle must not be collected while the graph can replay. Every *_device call made
# while the stream is capturing puts its engine here; release_captured() (per engine or for all) lets go again.
_CAPTURED: set = set()


def _pin_if_capturing(engine, stream) -> None:
    import torch

    if torch.cuda.is_current_stream_capturing() or (stream is not None and hasattr(stream, "is_capturing")
                                                      and stream.is_capturing()):
        _CAPTURED.add(engine)


def release_captured(engine=None) -> int:
    """The caller states that the graphs which captured ``engine`` (default: every captured engine) are gone: the
    engines may be collected, and the estimator's scratch may grow, again. Returns how many engines were released.
    Only ``release_captured()`` without an argument -- "no graph of this process will replay any more" -- also frees the
    engines that were destroyed while pinned (the library's parked list is process-wide)."""
    todo = list(_CAPTURED) if engine is None else [e for e in (engine,) if e in _CAPTURED]
    for e in todo:
        _CAPTURED.discard(e)
        e._release_graphs()
    # mof_purge_deferred() frees EVERY parked engine of the process -- also engines that were closed while a graph other than
    # the caller's still replays through them (include/mof.h: purge when the graphs are destroyed). Releasing ONE engine says
    # nothing about those, so only the all-engines form ("every graph is gone") purges.
    if engine is None:
        _capi.load().mof_purge_deferred()
    return len(todo)


PEAK_OPENCV, PEAK_OCL = 0, 1  # include/mof.h
INTER_CUBIC, INTER_LANCZOS4 = 2, 4  # include/mof.h (cv::INTER_CUBIC, cv::INTER_LANCZOS4)
LOGPOLAR_CV4, LOGPOLAR_CV3 = 0, 1  # include/mof.h


def _check_device_batch(cur, prev, frame_hw, device_index: int, channels: int = 1) -> None:
    """The C ABI receives raw pointers, a pitch and a frame stride: everything it cannot see is checked here.
    cur/prev: torch uint8 [n, H, W] (or [n, H, W, 3]) on the engine's device, innermost dimension(s) dense."""
    import torch

    want_dim = 3 if channels == 1 else 4
    for name, t in (("cur", cur), ("prev", prev)):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.uint8:
            raise ValueError(f"{name} must be a torch uint8 tensor")
        if not t.is_cuda:
            raise ValueError(f"{name} must live on the GPU (there is no CPU path)")
        if t.device.index != device_index:
            raise ValueError(f"{name} is on cuda:{t.device.index}, the engine on cuda:{device_index}")
        if t.dim() != want_dim:
            raise ValueError(f"{name} must have {want_dim} dimensions, got {t.dim()}")
        if tuple(t.shape[1:3]) != tuple(frame_hw):
            raise ValueError(f"{name} frames are {tuple(t.shape[1:3])}, engine expects {tuple(frame_hw)}")
        if channels == 3 and (t.shape[3] != 3 or t.stride(3) != 1 or t.stride(2) != 3):
            raise ValueError(f"{name} must be interleaved BGR8 ([n, H, W, 3], pixel stride 3)")
        if channels == 1 and t.stride(2) != 1:
            raise ValueError(f"{name} rows must be dense (stride 1 along x)")
        if t.stride(1) < t.shape[2] * channels or t.stride(0) < 0:
            raise ValueError(f"{name} has overlapping rows or a negative frame stride")
    if cur.shape != prev.shape:
        raise ValueError(f"cur {tuple(cur.shape)} and prev {tuple(prev.shape)} differ")
    if cur.stride(1) != prev.stride(1):
        raise ValueError("cur and prev must share one row pitch")


class FftMethod:
    """FftMethod behind the C ABI. ``layout`` generalises the reference's square tiling
    (origin, stride, grid); without it the constructor normalises the geometry exactly
    as /root/reference/src/FftMethod.cpp:1706-1720 does. ``peak_model`` selects which of the reference's two
    peak models runs (include/mof.h): PEAK_OPENCV = cv::phaseCorrelate's (useOCL=false, the default and the path
    BASELINE.json names), PEAK_OCL = its OpenCL kernel's (useOCL=true; ``search_radius`` = SEARCH_RADIUS, 55)."""

    def __init__(self, frame_size: int | None = None, sample_point_size: int = 64, max_px_speed: float = 80.0, *,
                 frame_shape: tuple[int, int] | None = None, grid: tuple[int, int] | None = None,
                 origin: tuple[int, int] = (0, 0), stride: tuple[int, int] | None = None, device: int = 0,
                 peak_model: int = 0, search_radius: int = 55):
        lib = _capi.load()
        cfg = FftConfig()
        if frame_shape is None:
            if frame_size is None:
                raise ValueError("frame_size or frame_shape required")
            check(lib.mof_fft_config_reference(C.byref(cfg), frame_size, sample_point_size, float(max_px_speed)))
        else:
            h, w = frame_shape
            stride = stride or (sample_point_size, sample_point_size)
            grid = grid or ((w - origin[0] - sample_point_size) // stride[0] + 1,
                            (h - origin[1] - sample_point_size) // stride[1] + 1)
            cfg = FftConfig(w, h, sample_point_size, grid[0], grid[1], origin[0], origin[1], stride[0], stride[1],
                            float(max_px_speed), 0, 0, 55)
        cfg.device = device
        cfg.peak_model = int(peak_model)
        cfg.search_radius = int(search_radius)
        self.cfg = cfg
        self._lib = lib
        self._h = C.c_void_p()
        check(lib.mof_fft_create(C.byref(cfg), C.byref(self._h)))

    @property
    def kernel_variant(self) -> str:
        """'stockham' or 'quad' (diagnostics, include/mof.h)."""
        return self._lib.mof_fft_kernel_variant(self._h).decode()

    # -- reference surface --------------------------------------------------------------------
    @property
    def sqNum(self) -> int:
        return self.cfg.grid_x

    @property
    def n_patches(self) -> int:
        return self.cfg.grid_x * self.cfg.grid_y

    def setImPrev(self, frame) -> None:
        f = _np_u8(frame)
        self._check_shape(f)
        check(self._lib.mof_fft_set_prev(self._h, f.ctypes.data, f.strides[0]))

    def reset(self) -> None:
        check(self._lib.mof_fft_reset(self._h))

    def processImage(self, imCurr, gui=False, debug=False, midPoint=None, yaw_angle=0.0, rot_center=None,
                     raw_output=None, fx=300.0, fy=300.0) -> np.ndarray:
        """Returns [grid_y*grid_x, 2] float64 shifts, index i + j*grid_x, NaN = invalid.
        The extra arguments are accepted and ignored, as FftMethod ignores them."""
        f = _np_u8(imCurr)
        self._check_shape(f)
        out = np.empty((self.n_patches, 2), np.float64)
        ninv = C.c_int(0)
        check(self._lib.mof_fft_process(self._h, f.ctypes.data, f.strides[0], out.ctypes.data, C.byref(ninv)))
        self.last_invalid = ninv.value
        return out

    def processImageLongRange(self, imCurr, gui=False, debug=False, midPoint=None, yaw_angle=0.0, rot_center=None,
                              raw_output=None, fx=300.0, fy=300.0) -> np.ndarray:
        """FftMethod::processImageLongRange: [sqNum_lr^2, 2] shifts in quarter-resolution pixels."""
        f = _np_u8(imCurr)
        self._check_shape(f)
        n = self._lib.mof_fft_long_range_patches(self._h)
        if n < 0:
            check(n)
        out = np.empty((n, 2), np.float64)
        ninv = C.c_int(0)
        check(self._lib.mof_fft_process_long_range(self._h, f.ctypes.data, f.strides[0], out.ctypes.data, C.byref(ninv)))
        self.last_invalid = ninv.value
        return out

    def process_long_range_batch_device(self, cur, prev, stream=None):
        import torch

        _check_device_batch(cur, prev, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device)
        n_lr = self._lib.mof_fft_long_range_patches(self._h)
        if n_lr < 0:
            check(n_lr)
        n = cur.shape[0]
        out = torch.empty((n, n_lr, 2), dtype=torch.float64, device=cur.device)
        s = stream if stream is not None else torch.cuda.current_stream(cur.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_fft_process_long_range_batch_device(self._h, cur.data_ptr(), cur.stride(0), prev.data_ptr(),
                                                                prev.stride(0), cur.stride(1), n, out.data_ptr(),
                                                                _stream_ptr(s)))
        return out

    # -- batched --------------------------------------------------------------------------------
    def process_batch_host(self, cur: np.ndarray, prev: np.ndarray) -> np.ndarray:
        cur, prev, cs, ps, pitch = _host_batch_views(cur, prev)
        self._check_shape(cur)
        n = cur.shape[0]
        out = np.empty((n, self.n_patches, 2), np.float64)
        check(self._lib.mof_fft_process_batch_host(self._h, cur.ctypes.data, cs, prev.ctypes.data, ps, pitch, n, out.ctypes.data))
        return out

    def process_batch_device(self, cur, prev, out=None, stream=None):
        """cur, prev: torch uint8 tensors [n, H, W] on this engine's device (last dim contiguous,
        any row pitch / frame stride). Asynchronous on torch's current stream. Returns a
        float64 tensor [n, patches, 2]."""
        import torch

        _check_device_batch(cur, prev, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device)
        n = cur.shape[0]
        if out is None:
            out = torch.empty((n, self.n_patches, 2), dtype=torch.float64, device=cur.device)
        if not (out.is_cuda and out.device == cur.device and out.is_contiguous() and out.dtype == torch.float64
                and out.numel() == n * self.n_patches * 2):
            raise ValueError("out must be a dense float64 tensor of n * patches * 2 elements on the engine's device")
        s = stream if stream is not None else torch.cuda.current_stream(cur.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_fft_process_batch_device(self._h, cur.data_ptr(), cur.stride(0), prev.data_ptr(),
                                                     prev.stride(0), cur.stride(1), n, out.data_ptr(), _stream_ptr(s)))
        return out

    def process_sequence_device(self, frames, out=None, stream=None):
        """frames: torch uint8 [n, H, W] video on this engine's device -> float64 [n - 1, patches, 2]: pair k = (frame k + 1,
        frame k), what consecutive processImage calls return after the first (FftMethod.cpp:1872). Asynchronous."""
        import torch

        _check_device_batch(frames, frames, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device)
        n = frames.shape[0]
        if out is None:
            out = torch.empty((max(n - 1, 0), self.n_patches, 2), dtype=torch.float64, device=frames.device)
        if not (out.is_cuda and out.device == frames.device and out.is_contiguous() and out.dtype == torch.float64
                and out.numel() == max(n - 1, 0) * self.n_patches * 2):
            raise ValueError("out must be a dense float64 tensor of (n - 1) * patches * 2 elements on the engine's device")
        s = stream if stream is not None else torch.cuda.current_stream(frames.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_fft_process_sequence_device(self._h, frames.data_ptr(), frames.stride(0), frames.stride(1), n,
                                                        out.data_ptr(), _stream_ptr(s)))
        return out

    def process_sequence_device_bgr(self, frames, stream=None):
        """frames: torch uint8 [n, H, W, 3] BGR8 video (W-stride 3, any row pitch) -> float64 [n - 1, patches, 2]; CV_RGB2GRAY
        fused into the sequence kernels' loads, identical bits to process_sequence_device on the converted frames."""
        import torch

        _check_device_batch(frames, frames, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device, channels=3)
        n = frames.shape[0]
        out = torch.empty((max(n - 1, 0), self.n_patches, 2), dtype=torch.float64, device=frames.device)
        s = stream if stream is not None else torch.cuda.current_stream(frames.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_fft_process_sequence_device_bgr(self._h, frames.data_ptr(), frames.stride(0), frames.stride(1), n,
                                                            out.data_ptr(), _stream_ptr(s)))
        return out

    def process_batch_device_bgr(self, cur, prev, stream=None):
        """cur, prev: torch uint8 [n, H, W, 3] BGR8 views (crop of the camera frames; W-stride 3, any row pitch):
        CV_RGB2GRAY (as the node applies it to BGR data) is fused into the kernel's load."""
        import torch

        _check_device_batch(cur, prev, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device, channels=3)
        n = cur.shape[0]
        out = torch.empty((n, self.n_patches, 2), dtype=torch.float64, device=cur.device)
        s = stream if stream is not None else torch.cuda.current_stream(cur.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_fft_process_batch_device_bgr(self._h, cur.data_ptr(), cur.stride(0), prev.data_ptr(),
                                                         prev.stride(0), cur.stride(1), n, out.data_ptr(),
                                                         _stream_ptr(s)))
        return out

    def _check_shape(self, f) -> None:
        if tuple(f.shape[-2:]) != (self.cfg.frame_height, self.cfg.frame_width):
            raise ValueError(f"frame is {tuple(f.shape[-2:])}, engine expects "
                             f"{(self.cfg.frame_height, self.cfg.frame_width)}")

    def _release_graphs(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            check(self._lib.mof_fft_release_graphs(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mof_fft_destroy(self._h)  # deferred by the library while a captured graph pins the engine
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BmBase:
    def __init__(self, cfg: BmConfig, device: int):
        self._lib = _capi.load()
        cfg.device = device
        self.cfg = cfg
        self._h = C.c_void_p()
        check(self._lib.mof_bm_create(C.byref(cfg), C.byref(self._h)))

    @property
    def n_blocks(self) -> int:
        return self.cfg.grid_x * self.cfg.grid_y

    def _check_shape(self, f) -> None:
        if tuple(f.shape[-2:]) != (self.cfg.frame_height, self.cfg.frame_width):
            raise ValueError(f"frame is {tuple(f.shape[-2:])}, engine expects "
                             f"{(self.cfg.frame_height, self.cfg.frame_width)}")

    def setImPrev(self, frame) -> None:
        f = _np_u8(frame)
        self._check_shape(f)
        check(self._lib.mof_bm_set_prev(self._h, f.ctypes.data, f.strides[0]))

    def reset(self) -> None:
        check(self._lib.mof_bm_reset(self._h))

    def processBlocks(self, imCurr):
        """Integer stage: (dx[gy,gx], dy[gy,gx], (modeX, modeY))."""
        f = _np_u8(imCurr)
        self._check_shape(f)
        dx = np.empty(self.n_blocks, np.int8)
        dy = np.empty(self.n_blocks, np.int8)
        mode = np.zeros(2, np.int8)
        check(self._lib.mof_bm_process(self._h, f.ctypes.data, f.strides[0], dx.ctypes.data, dy.ctypes.data,
                                       mode.ctypes.data))
        g = (self.cfg.grid_y, self.cfg.grid_x)
        return dx.reshape(g), dy.reshape(g), (int(mode[0]), int(mode[1]))

    def refine(self, fullpix, passes: int = 2, faithful: bool = True):
        """BlockMethod::Refine on the frames of the last processImage/processBlocks call (BlockMethod.cpp:96-147)."""
        out = np.zeros(2, np.float64)
        check(self._lib.mof_bm_refine(self._h, int(fullpix[0]), int(fullpix[1]), passes, int(faithful), out.ctypes.data))
        return float(out[0]), float(out[1])

    def processImage(self, imCurr, gui=False, debug=False, midPoint=None, yaw_angle=0.0, tiltCorr=None, refine=None):
        """One output vector. refine=None: the histogram mode (FastSpacedBMMethod_OCL.cpp:172-175; BlockMethod before
        :79). refine="faithful"/"fixed": BlockMethod's full return value, Refine(mode, 2) (BlockMethod.cpp:79)."""
        _, _, mode = self.processBlocks(imCurr)
        if refine is None:
            return np.array([[float(mode[0]), float(mode[1])]])
        return np.array([self.refine(mode, 2, refine == "faithful")])

    def process_batch_host(self, cur: np.ndarray, prev: np.ndarray):
        cur, prev, cs, ps, pitch = _host_batch_views(cur, prev)
        self._check_shape(cur)
        n = cur.shape[0]
        dx = np.empty((n, self.cfg.grid_y, self.cfg.grid_x), np.int8)
        dy = np.empty_like(dx)
        mode = np.empty((n, 8), np.int8)
        check(self._lib.mof_bm_process_batch_host(self._h, cur.ctypes.data, cs, prev.ctypes.data, ps, pitch, n,
                                                  dx.ctypes.data, dy.ctypes.data, mode.ctypes.data))
        return dx, dy, mode

    def process_batch_device(self, cur, prev, stream=None):
        import torch

        _check_device_batch(cur, prev, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device)
        n = cur.shape[0]
        dx = torch.empty((n, self.cfg.grid_y, self.cfg.grid_x), dtype=torch.int8, device=cur.device)
        dy = torch.empty_like(dx)
        mode = torch.empty((n, 8), dtype=torch.int8, device=cur.device)
        s = stream if stream is not None else torch.cuda.current_stream(cur.device)
        check(self._lib.mof_bm_process_batch_device(self._h, cur.data_ptr(), cur.stride(0), prev.data_ptr(),
                                                    prev.stride(0), cur.stride(1), n, dx.data_ptr(), dy.data_ptr(),
                                                    mode.data_ptr(), _stream_ptr(s)))
        return dx, dy, mode

    def process_batch_device_bgr(self, cur, prev, stream=None):
        """cur, prev: torch uint8 [n, H, W, 3] BGR8 views (W-stride 3, any row pitch): CV_RGB2GRAY (as the node applies it to
        BGR data) is fused into the kernels' staging loads; same bits as the gray entry on the converted frames."""
        import torch

        _check_device_batch(cur, prev, (self.cfg.frame_height, self.cfg.frame_width), self.cfg.device, channels=3)
        n = cur.shape[0]
        dx = torch.empty((n, self.cfg.grid_y, self.cfg.grid_x), dtype=torch.int8, device=cur.device)
        dy = torch.empty_like(dx)
        mode = torch.empty((n, 8), dtype=torch.int8, device=cur.device)
        s = stream if stream is not None else torch.cuda.current_stream(cur.device)
        check(self._lib.mof_bm_process_batch_device_bgr(self._h, cur.data_ptr(), cur.stride(0), prev.data_ptr(),
                                                        prev.stride(0), cur.stride(1), n, dx.data_ptr(), dy.data_ptr(),
                                                        mode.data_ptr(), _stream_ptr(s)))
        return dx, dy, mode

    def _release_graphs(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            check(self._lib.mof_bm_release_graphs(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mof_bm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlockMethod(_BmBase):
    """BlockMethod(frameSize, samplePointSize, scanRadius, ...) -- /root/reference/src/BlockMethod.cpp:3-22."""

    def __init__(self, frameSize: int, samplePointSize: int, scanRadius: int, scanDiameter: int | None = None,
                 scanCount: int | None = None, stepSize: int = 0, device: int = 0):
        cfg = BmConfig()
        check(_capi.load().mof_bm_config_block_method(C.byref(cfg), frameSize, samplePointSize, scanRadius))
        super().__init__(cfg, device)


class FastSpacedBMMethod(_BmBase):
    """FastSpacedBMMethod(samplePointSize, scanRadius, stepSize, ...) on a W x H frame --
    /root/reference/src/FastSpacedBMMethod_OCL.cpp:5-6, :81-97."""

    def __init__(self, samplePointSize: int, scanRadius: int, stepSize: int, frame_shape: tuple[int, int],
                 device: int = 0):
        cfg = BmConfig()
        h, w = frame_shape
        check(_capi.load().mof_bm_config_fast_spaced(C.byref(cfg), w, h, samplePointSize, stepSize, scanRadius))
        super().__init__(cfg, device)


class ScaleRotationEstimator:
    """scaleRotationEstimator(resolution, m, ...) -- /root/reference/src/scaleRotationEstimator.cpp:3-32; processImage
    returns (scale, rotation [rad]) like :34-148."""

    def __init__(self, resolution: int, m: float = 49.9, storeVideo: bool = False, videoPath=None, videoFPS: int = 30,
                 device: int = 0, logpolar_variant: int = 0, batch_chunk: int = 0, pipeline_lanes: int = 0):
        """logpolar_variant: LOGPOLAR_CV4 (cv::logPolar of OpenCV 4.x, ROS Noetic) or LOGPOLAR_CV3 (cvLogPolar of OpenCV 3.2,
        ROS Melodic) -- the two calls scaleRotationEstimator.cpp:41-46 compiles. batch_chunk / pipeline_lanes: batched mode
        only (frame pairs per pipeline pass, one or two stream lanes; 0 = the library's defaults, see mof.h)."""
        self._lib = _capi.load()
        self.cfg = SrConfig(resolution, float(m), device, int(logpolar_variant), int(batch_chunk), int(pipeline_lanes))
        self._h = C.c_void_p()
        check(self._lib.mof_sr_create(C.byref(self.cfg), C.byref(self._h)))

    def reset(self) -> None:
        check(self._lib.mof_sr_reset(self._h))

    def reserve(self, n_pairs: int) -> None:
        """Grow the batch scratch now (needed before the first batch of a fresh engine is captured into a HIP graph)."""
        check(self._lib.mof_sr_reserve(self._h, int(n_pairs)))

    def processImage(self, imCurr, gui=False, debug=False):
        f = _np_u8(imCurr)
        if f.shape != (self.cfg.resolution, self.cfg.resolution):
            raise ValueError("scaleRotationEstimator accepts only square images of its resolution")
        out = np.zeros(2, np.float64)
        check(self._lib.mof_sr_process(self._h, f.ctypes.data, f.strides[0], out.ctypes.data))
        return float(out[0]), float(out[1])

    def process_sequence_host(self, frames) -> np.ndarray:
        """frames: numpy uint8 [n, res, res] video in HOST memory (any frame stride / row pitch; pinned_empty() memory is DMA'd in place)
        -> float64 [n, 4] = what n consecutive processImage calls return (scale, rot, pt.x, pt.y), continuing and updating the engine's
        state; ``self.last_gated`` = gated frames. Synchronous (mof_sr_process_sequence_host)."""
        f = np.asarray(frames)
        if f.ndim != 3 or tuple(f.shape[1:]) != (self.cfg.resolution, self.cfg.resolution):
            raise ValueError("scaleRotationEstimator accepts only square images of its resolution")
        if not (f.dtype == np.uint8 and f.strides[2] == 1 and f.strides[1] >= f.shape[2] and (f.shape[0] < 2 or f.strides[0] > 0)):
            f = np.ascontiguousarray(f, dtype=np.uint8)
        out = np.zeros((f.shape[0], 4), np.float64)
        gated = C.c_int(0)
        check(self._lib.mof_sr_process_sequence_host(self._h, f.ctypes.data, max(f.strides[0], 0), f.strides[1], f.shape[0], out.ctypes.data,
                                                     C.byref(gated)))
        self.last_gated = gated.value
        return out

    def process_batch_device(self, cur, prev, stream=None):
        """cur, prev: torch uint8 [n, res, res] views (any pitch/stride) -> float64 [n, 4] = scale, rot, pt.x, pt.y."""
        import torch

        _check_device_batch(cur, prev, (self.cfg.resolution, self.cfg.resolution), self.cfg.device)
        n = cur.shape[0]
        out = torch.empty((n, 4), dtype=torch.float64, device=cur.device)
        s = stream if stream is not None else torch.cuda.current_stream(cur.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_sr_process_batch_device(self._h, cur.data_ptr(), cur.stride(0), prev.data_ptr(),
                                                    prev.stride(0), cur.stride(1), n, out.data_ptr(), _stream_ptr(s)))
        return out

    def process_sequence_device(self, frames, resolve_gate: bool = True, stream=None):
        """frames: torch uint8 [n, res, res] view of a video (any pitch / frame stride) -> float64 [n, 4] = what n consecutive
        processImage calls return (scale, rot, pt.x, pt.y; the first frame of a fresh estimator: 1, 0, 0, 0), continuing
        and updating the engine's stateful sequence. resolve_gate=True is synchronous and exact under the gate of
        scaleRotationEstimator.cpp:119-121 (``self.last_gated`` = gated frames); False is asynchronous / capturable."""
        import torch

        _check_device_batch(frames, frames, (self.cfg.resolution, self.cfg.resolution), self.cfg.device)
        n = frames.shape[0]
        out = torch.empty((n, 4), dtype=torch.float64, device=frames.device)
        s = stream if stream is not None else torch.cuda.current_stream(frames.device)
        _pin_if_capturing(self, s)
        gated = C.c_int(0)
        check(self._lib.mof_sr_process_sequence_device(self._h, frames.data_ptr(), frames.stride(0), frames.stride(1), n,
                                                       out.data_ptr(), _stream_ptr(s),
                                                       C.byref(gated) if resolve_gate else None))
        self.last_gated = gated.value if resolve_gate else None
        return out

    def logpolar_batch_device(self, src, interpolation: int = INTER_LANCZOS4, dst=None, stream=None):
        """cv::logPolar of n res x res crops (torch uint8 [n, res, res], any pitch/stride) -> uint8 [n, res, res].
        Destination pixels whose source lies outside the image keep the content of ``dst`` (zeros when omitted)."""
        import torch

        _check_device_batch(src, src, (self.cfg.resolution, self.cfg.resolution), self.cfg.device)
        n, res = src.shape[0], self.cfg.resolution
        if dst is None:
            dst = torch.zeros((n, res, res), dtype=torch.uint8, device=src.device)
        if not (dst.is_cuda and dst.device == src.device and dst.dtype == torch.uint8 and dst.is_contiguous()
                and tuple(dst.shape) == (n, res, res)):
            raise ValueError("dst must be a dense uint8 [n, res, res] tensor on the engine's device")
        s = stream if stream is not None else torch.cuda.current_stream(src.device)
        _pin_if_capturing(self, s)
        check(self._lib.mof_sr_logpolar_batch_device(self._h, src.data_ptr(), src.stride(0), src.stride(1), n,
                                                     int(interpolation), dst.data_ptr(), _stream_ptr(s)))
        return dst

    def _release_graphs(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            check(self._lib.mof_sr_release_graphs(self._h))

    @property
    def graph_pinned(self) -> bool:
        return bool(self._lib.mof_sr_graph_pinned(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mof_sr_destroy(self._h)  # deferred by the library while a captured graph pins the engine
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
