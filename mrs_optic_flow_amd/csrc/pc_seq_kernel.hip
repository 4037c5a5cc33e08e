// pc_seq_kernel.hip -- K1 for FRAME SEQUENCES (64 x 64 patches) on gfx950.
//
// FftMethod::processImage keeps the current frame as the next call's previous one (`imPrev = imCurr.clone()`,
// /root/reference/src/FftMethod.cpp:1872): in a video every frame is `cur` of one patch pair and `prev` of the next.
// The pair kernel (pc_kernel.hip) packs cur + i*prev into one complex transform -- the cheapest form for independent
// pairs (1 complex 2-D transform forward + 1/2 back) -- but on a video it transforms every frame twice. Here a
// workgroup owns one PATCH POSITION and walks a run of consecutive frames: per new frame ONE real 2-D transform forward
// (half a complex one) and the Hermitian inverse (another half): 1.0 instead of 1.5 units per pair, half the pixel
// conversions, and the previous frame's half spectrum never leaves the registers (8 complex bins per lane).
//
// Per frame, N = 64, 256 lanes, the same skewed 64 x 72 LDS tile as K1 (pc_passes.hpp):
//   1. the wave's 16 rows are stored as 8 complex lines (rows 2j, 2j+1 = real, imaginary part), transformed along x
//      (row_pass) and untangled into the rows' half spectra u = 0..31 (column 0 carries the real u = 0 and u = 32 bins)
//      -- all wave-local, no workgroup barrier;
//   2. barrier; the wave's 8 columns go forward along y, meet the previous frame's spectrum held in registers in the
//      normalised cross-power spectrum (same rules as K1: pc_common.hpp), and go straight on into the inverse column
//      transform -- 64 = 8 x 8, so the output distribution of a forward second stage IS the input distribution of the
//      next first stage: no LDS round trip in between;
//   3. barrier; row PAIRS (y, y + 32) ride one complex transform along x (the surface is real); arg-max (fftShift +
//      first maximum in row-major order) from the registers of its last stage;
//   4. two barriers around the 5 x 5 centroid + gate of wave 0, as K1.
// Four barriers per frame (K1: five per pair). Results: the same estimator as K1 on (frames[k+1], frames[k]) -- same
// arg-max, sub-pixel shifts equal within rounding (1e-4 px bar against the oracle, tests/test_gpu_fft_sequence.py).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"
#include "pc_passes.hpp"

namespace mof {

namespace {

constexpr int SQN = 64, SQH = 32;

// Step 2 for the wave's 8 columns [col0, col0 + 8): forward along y, cross-power against `prev`, inverse along y.
// prime: the first frame of a run only leaves its spectrum in the registers.
template <int PK>
__device__ __forceinline__ void col_pass_fused64(cf* __restrict__ z, int col0, int lane, const cf* tw_col, cf* prev, cf& prev0,
                                                 cf& prevH, bool prime, bool has_col0) {
  constexpr int N = SQN, H = SQH;
  const int col = col0 + (lane & 7), x = lane >> 3;
  cf v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds_read(&z[zaddr<N>(x + 8 * k, col)]);
  butterfly<8>(v);
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) z[zaddr<N>(x * 8 + k, col)] = v[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds_read(&z[zaddr<N>(x + 8 * k, col)]);
  butterfly8_tw(v, tw_col);  // v[k] = 2 F[x + 8k][col]: the new frame's spectrum (doubled, as everything here)
  cf C0 = {0.f, 0.f}, Ch = {0.f, 0.f};
  if (has_col0) {
    // column 0 is the packed pair of the two REAL columns u = 0 and u = H: G[v] = F[v][0] + i F[v][H]; both are
    // Hermitian along v, so they come apart with the partner bin N - v -- which another lane holds: through LDS
    wave_sync();
    if ((lane & 7) == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) z[zaddr<N>(x + 8 * k, 0)] = v[k];
    }
    wave_sync();
    const int vv = lane, vm = (N - lane) & (N - 1);
    const bool mine = lane <= H, self = lane == 0 || lane == H;
    if (mine) {
      cf f0, fh;
      untangle2(lds_read(&z[zaddr<N>(vv, 0)]), lds_read(&z[zaddr<N>(vm, 0)]), &f0, &fh);
      f0 = {0.5f * f0.x, 0.5f * f0.y};  // (the untangle doubles once more)
      fh = {0.5f * fh.x, 0.5f * fh.y};
      if (!prime) {
        C0 = cross_power_ab<PK>(f0, prev0, self);  // the four real-only slots are (0|H, 0|H)
        Ch = cross_power_ab<PK>(fh, prevH, self);
      }
      prev0 = f0;
      prevH = fh;
    }
    wave_sync();
    if (mine && !prime) {  // conj(C[v][0]) + i conj(C[v][H]) for v and N - v
      z[zaddr<N>(vv, 0)] = {C0.x + Ch.y, Ch.x - C0.y};
      if (!self) z[zaddr<N>(vm, 0)] = {C0.x - Ch.y, Ch.x + C0.y};
    }
    wave_sync();
  }
  if (prime) {
#pragma unroll
    for (int k = 0; k < 8; ++k) prev[k] = v[k];
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cf C = cross_power_ab<PK>(v[k], prev[k], false);
    prev[k] = v[k];
    v[k] = {C.x, -C.y};
  }
  if (has_col0 && (lane & 7) == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lds_read(&z[zaddr<N>(x + 8 * k, 0)]);
  }
  // inverse along y = forward transform of conj(C): the registers already hold rows x + 8k of the column
  butterfly<8>(v);
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) z[zaddr<N>(x * 8 + k, col)] = v[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds_read(&z[zaddr<N>(x + 8 * k, col)]);
  butterfly8_tw(v, tw_col);
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) z[zaddr<N>(x + 8 * k, col)] = v[k];
  wave_sync();
}

// Step 3 for the wave's 8 row pairs (y1 = row0 + lane / 8, y1 + 32): columns 0..31 of both rows hold F1[y][u] (column 0:
// the real F1[y][0], F1[y][32]); F1[y][64 - u] = conj F1[y][u]. Output z(y1, x) = (c[y1][x], c[y1 + 32][x]).
template <int PK>
__device__ __forceinline__ Best row_pass_inv64(cf* __restrict__ z, int row0, int lane, const cf* tw_row, int search_radius) {
  constexpr int N = SQN, H = SQH;
  const int y1 = row0 + (lane >> 3), y2 = y1 + H, x = lane & 7;
  const bool x0 = x == 0;
  cf v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int u = x + 8 * k;
    const int uu = (k < 4) ? u : ((k == 4 && x0) ? 0 : N - u);
    const cf a = lds_read(&z[zaddr<N>(y1, uu)]), c = lds_read(&z[zaddr<N>(y2, uu)]);
    cf e;
    if (k < 4) {
      e = {a.x - c.y, a.y + c.x};
      if (k == 0 && x0) e = {a.x, c.x};
    } else {
      e = {a.x + c.y, c.x - a.y};
      if (k == 4 && x0) e = {a.y, c.y};
    }
    v[k] = e;
  }
  butterfly<8>(v);
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) z[zaddr<N>(y1, x * 8 + k)] = v[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds_read(&z[zaddr<N>(y1, x + 8 * k)]);
  butterfly8_tw(v, tw_row);
  if constexpr (PK == 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k].x = ocl_scale_mask<N>(v[k].x, y1, x + 8 * k, search_radius);
      v[k].y = ocl_scale_mask<N>(v[k].y, y2, x + 8 * k, search_radius);
    }
  }
  wave_sync();
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int k = 0; k < 8; ++k) m = fmaxf(m, fmaxf(v[k].x, v[k].y));
  int mi = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int xx = x + 8 * k, xs = (xx + H) & (N - 1);
    z[zaddr<N>(y1, xx)] = v[k];
    mi = min(mi, v[k].x == m ? y2 * N + xs : 0x7fffffff);  // row y1      -> shifted row y1 + 32
    mi = min(mi, v[k].y == m ? y1 * N + xs : 0x7fffffff);  // row y1 + 32 -> shifted row y1
  }
  return Best{m, mi};
}

// CH = 3: the frames are interleaved BGR8 and CV_RGB2GRAY (as the node applies it, optic_flow.cpp:1622) happens in the load
template <int PK, int CH>
__global__ void __launch_bounds__(256) pc_seq_kernel(PcArgs a, int n_pairs, int run) {
  constexpr int N = SQN, H = SQH;
  using P = PcTraits<N>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  Best* red = reinterpret_cast<Best*>(z + P::TILE);
  const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
  const int p0 = blockIdx.z * run;                          // first pair of this run: frames p0 .. p0 + np
  const int np = n_pairs - p0 < run ? n_pairs - p0 : run;
  const int patches = a.grid_x * a.grid_y, patch = blockIdx.y * a.grid_x + blockIdx.x;
  const int px0 = a.origin_x + blockIdx.x * a.stride_x, py0 = a.origin_y + blockIdx.y * a.stride_y;
  // this lane's 2 x 8 pixels: rows 2j, 2j + 1 of the patch (j = 8 wave + lane / 8), columns 8 (lane % 8) .. +7
  const uint8_t* src = a.cur + (size_t)py0 * a.pitch + CH * px0 + (size_t)(16 * wave0 + 2 * (lane0 >> 3)) * a.pitch + CH * 8 * (lane0 & 7);
  cf tw_row[7], tw_col[7];
  {
    const int xr = lane0 & 7, xc = lane0 >> 3;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      tw_row[k - 1] = {a.twiddles[2 * (k * xr)], a.twiddles[2 * (k * xr) + 1]};
      tw_col[k - 1] = {a.twiddles[2 * (k * xc)], a.twiddles[2 * (k * xc) + 1]};
    }
  }
  cf prev[8], prev0 = {0.f, 0.f}, prevH = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) prev[k] = {0.f, 0.f};
  uint32_t ra[2], rb[2];  // the next frame's pixels, requested one frame ahead
  auto fetch = [&](int f) {
    const uint8_t* s = src + (size_t)f * a.cur_stride;
    if constexpr (CH == 1) {
      __builtin_memcpy(ra, s, 8);
      __builtin_memcpy(rb, s + a.pitch, 8);
    } else {
      gray8_from_bgr24(s, ra);
      gray8_from_bgr24(s + a.pitch, rb);
    }
  };
  fetch(p0);
  for (int f = 0; f <= np; ++f) {  // frame p0 + f; f = 0 primes the registers, f >= 1 closes pair p0 + f - 1
    // (lane / wave laundered once per frame: keeps LICM from hoisting every LDS address out of the loop, see K1)
    int lane = lane0, wave = wave0;
    asm volatile("" : "+v"(lane), "+v"(wave));
    lane &= 63;
    wave &= 3;
    // ---- 1. the wave's 8 complex lines (tile rows 16 wave .. +7): u8 -> f32 (convertTo, :1805-1806), row transforms
    {
      const int lr = 16 * wave + (lane >> 3);
#if MOF_RAW_STAGE  // raw pixel staging (pc_passes.hpp): one ds_write_b128 instead of eight ds_write_b64 of converted pixels
      raw_store8<N>(z, 16 * wave, lane, ra, rb);
#else
      const int c0 = 8 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        z[zaddr<N>(lr, c0 + i)] = {(float)((ra[i >> 2] >> (8 * (i & 3))) & 0xffu), (float)((rb[i >> 2] >> (8 * (i & 3))) & 0xffu)};
#endif
      if (f < np) fetch(p0 + f + 1);
      wave_sync();
      row_pass<N, 8, MOF_RAW_STAGE != 0>(z, 16 * wave, lane, tw_row);
      // untangle line j into rows 2j, 2j + 1 (doubled): lane = (line, u mod 8), u = ug + 8 m
      const int ug = lane & 7;
      cf zk[4], zm[4], z32 = {0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int u = ug + 8 * m;
        zk[m] = lds_read(&z[zaddr<N>(lr, u)]);
        zm[m] = lds_read(&z[zaddr<N>(lr, (N - u) & (N - 1))]);
      }
      if (ug == 0) z32 = lds_read(&z[zaddr<N>(lr, H)]);
      wave_sync();
      const int r0 = 16 * wave + 2 * (lane >> 3);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        cf A, B;
        untangle2(zk[m], zm[m], &A, &B);
        if (m == 0 && ug == 0) {  // u = 0 and u = 32 are real: they share column 0
          z[zaddr<N>(r0, 0)] = {A.x, 2.f * z32.x};
          z[zaddr<N>(r0 + 1, 0)] = {B.x, 2.f * z32.y};
        } else {
          z[zaddr<N>(r0, ug + 8 * m)] = A;
          z[zaddr<N>(r0 + 1, ug + 8 * m)] = B;
        }
      }
    }
    __syncthreads();
    // ---- 2. columns 8 wave .. +7: forward, cross-power against the previous frame, inverse (dft, mulSpectrums,
    //         magSpectrums, divSpectrums, idft: FftMethod.cpp:1491-1497)
    col_pass_fused64<PK>(z, 8 * wave, lane, tw_col, prev, prev0, prevH, f == 0, wave == 0);
    if (f == 0) {
      __syncthreads();  // the tile is rewritten by the next frame's lines
      continue;
    }
    __syncthreads();
    // ---- 3. row pairs + arg-max (fftShift :1297-1305, minMaxLoc :1539)
    Best best = row_pass_inv64<PK>(z, 8 * wave, lane, tw_row, a.search_radius);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
      best = better(best, o);
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    // ---- 4. 5x5 weighted centroid in double + validity gate (:1337-1383, :1838-1856), wave 0
    float wval = 0.f;
    if (wave == 0) {
      for (int w = 1; w < 4; ++w) best = better(best, red[w]);
      wval = centroid_window_value<N, PK>(best, lane, [&](int ys, int xs) {
        const int y = (ys + H) & (N - 1), x = (xs + H) & (N - 1);  // un-shifted position
        const cf s = z[zaddr<N>(y & (H - 1), x)];
        return y < H ? s.x : s.y;
      });
    }
    __syncthreads();
    if (wave == 0)
      centroid_gate_store<N, PK>(best, wval, lane, a.max_px_speed_sq, a.out + 2 * ((size_t)(p0 + f - 1) * patches + patch));
  }
}

}  // namespace

bool pc_sequence_supported(int patch_size) { return patch_size == 64; }

// Diagnostic knob (co-scheduling experiments): MOF_PC_EXTRA_LDS=<bytes> pads the dynamic LDS request, i.e. caps the
// workgroups per CU (as for K1, pc_kernel.hip)
static size_t seq_extra_lds() {
  static const size_t v = [] {
    const char* e = getenv("MOF_PC_EXTRA_LDS");
    return e ? (size_t)atol(e) : (size_t)0;
  }();
  return v;
}


hipError_t pc_configure_sequence() {
  const int lds = (int)(PcTraits<64>::LDS_BYTES + seq_extra_lds());
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_kernel<0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_kernel<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// a.cur = frame 0, a.cur_stride = bytes between frames; pair k = (frame k + 1, frame k), k < n_pairs; a.prev unused.
hipError_t launch_pc_sequence(const PcArgs& a, int n_pairs, int run, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  if (run == 0) {
    // r06 (tools/video_probe.py: 128 pairs of 8 x 8 patches ran slower as a video than as pairs): fixed runs of 16 pairs left half of the
    // resident slots empty on a short video. As the half-tile kernel does (pc_half_kernel.hip): workgroups of a launch all last run + ~0.5
    // transforms (a run's first frame has no inverse), the launch lasts ceil(workgroups / slots) rounds of that -- the least product wins.
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long slots = (long)cus * 4, patches = (long)a.grid_x * a.grid_y;  // four workgroups per CU (one 36 KB tile each)
    auto cost = [&](int r) {
      const long wgs = patches * ((n_pairs + r - 1) / r), rounds = (wgs + slots - 1) / slots;
      return (double)rounds * ((double)r + 0.5);
    };
    double best = cost(2);
    for (int r = 3; r <= 64; ++r) best = cost(r) < best ? cost(r) : best;
    // ... the SHORTEST run within 3 % of it: the dispatcher is not round-synchronous, long runs leave a ragged tail (c2seq, 1024 pairs: the
    // model rates runs of 64 1.5 % ahead of 16, measured they are 5 % behind)
    for (int r = 2; r <= 64 && run == 0; ++r)
      if (cost(r) <= 1.03 * best) run = r;
  }
  if (run < 1) run = 1;
  const int runs = (n_pairs + run - 1) / run;
  if (runs > 65535 || (a.channels != 1 && a.channels != 3) || a.downscale != 1) return hipErrorInvalidValue;
  const dim3 g((unsigned)a.grid_x, (unsigned)a.grid_y, (unsigned)runs);
  const size_t lds = PcTraits<64>::LDS_BYTES + seq_extra_lds();
  if (a.channels == 3) {
    if (a.peak_model == 1) hipLaunchKernelGGL((pc_seq_kernel<1, 3>), g, dim3(256), lds, stream, a, n_pairs, run);
    else hipLaunchKernelGGL((pc_seq_kernel<0, 3>), g, dim3(256), lds, stream, a, n_pairs, run);
  } else if (a.peak_model == 1) {
    hipLaunchKernelGGL((pc_seq_kernel<1, 1>), g, dim3(256), lds, stream, a, n_pairs, run);
  } else {
    hipLaunchKernelGGL((pc_seq_kernel<0, 1>), g, dim3(256), lds, stream, a, n_pairs, run);
  }
  return hipGetLastError();
}

}  // namespace mof
