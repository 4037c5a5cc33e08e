// mfma_rowdft.hip -- the MFMA experiment of SURVEY.md section 7 / the round-1 review (item 6), as a micro-benchmark.
//
// Question: is the FORWARD ROW PASS of K1 (N = 64) -- load two u8 patches, convert, 64 row DFTs of cur and prev, row
// spectra into the LDS tile -- faster as a dense product on the matrix cores than as Stockham stages on the VALU?
//
//   VALU form (what K1 does): z = cur + i prev packed, 64 complex 64-point transforms = two radix-8 stages in LDS.
//   MFMA form: the row DFT of a real row is x[64] . [C | S](64 x 64): 33 cosine columns (u = 0..32) and 31 sine columns
//     (u = 1..31) -- exactly 64 -- so per patch pair  [cur; prev](128 x 64) . W(64 x 64) with v_mfma_f32_32x32x16_f16:
//     u8 pixels are exact in f16; the twiddles are split W = W_hi + W_lo (two f16 products, f32 accumulation) to keep
//     ~2^-22 relative accuracy. 4 (M tiles) x 2 (N tiles) x 4 (K steps) x 2 (hi, lo) = 64 MFMA of 32 cycles per
//     patch pair = 2.1 MFLOP of "useful" work, 4.2 issued.
// Both kernels share the skeleton: one workgroup (256 threads) per patch pair, same HBM loads, row spectra written to
// LDS, read back once into a checksum (so nothing is dead code). The program validates both against a double-precision
// DFT on the host and prints times for `n` patch pairs (default 65,536 = BASELINE c2's 1024 x 8 x 8).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I mrs_optic_flow_amd/csrc -I include \
//            -o gpurun_out/mfma_rowdft tools/ubench/mfma_rowdft.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pc_common.hpp"

using namespace mof;

namespace {

constexpr int N = 64, PITCH = 72;  // K1's tile: element c of a row at c + (c >> 3), pitch 72 complex
__device__ __forceinline__ int zaddr(int r, int c) { return r * PITCH + c + (c >> 3); }

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

// ---- VALU form: K1's load + row pass (two radix-8 Stockham stages, wave-local) ---------------------------------
__global__ void __launch_bounds__(256) rows_valu(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                 const float* __restrict__ tw, float* __restrict__ sums,
                                                 float2* __restrict__ dump, int n_dump) {
  __shared__ cf z[N * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, patch = blockIdx.x;
  const uint8_t* c = cur + (size_t)patch * N * N;
  const uint8_t* p = prev + (size_t)patch * N * N;
  {  // 16 px of cur and prev per lane (K1's load), packed as cur + i prev
    const int r = tid >> 2, x0 = (tid & 3) * 16;
    uint32_t a[4], b[4];
    __builtin_memcpy(a, c + r * N + x0, 16);
    __builtin_memcpy(b, p + r * N + x0, 16);
#pragma unroll
    for (int q = 0; q < 16; ++q)
      z[zaddr(r, x0 + q)] = {(float)((a[q >> 2] >> (8 * (q & 3))) & 0xffu), (float)((b[q >> 2] >> (8 * (q & 3))) & 0xffu)};
  }
  __syncthreads();
  // wave w owns rows 16w .. 16w+15; 64 = 8 x 8: stage 1 radix 8 (stride 8), stage 2 radix 8 with W_64^{k x}
  cf twr[7];
  {
    const int x = lane & 7;
#pragma unroll
    for (int k = 1; k < 8; ++k) twr[k - 1] = {tw[2 * (k * x)], tw[2 * (k * x) + 1]};
  }
  const int line0 = 16 * wave;
  {
    cf v[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, line = line0 + q / 8, x = q % 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[b][k] = lds_read(&z[zaddr(line, x + 8 * k)]);
      butterfly<8>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, line = line0 + q / 8, x = q % 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) z[zaddr(line, x * 8 + k)] = v[b][k];
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, line = line0 + q / 8, x = q % 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[b][k] = lds_read(&z[zaddr(line, x + 8 * k)]);
      butterfly8_tw(v[b], twr);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, line = line0 + q / 8, x = q % 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) z[zaddr(line, x + 8 * k)] = v[b][k];
    }
  }
  __syncthreads();
  // consumer: every thread reads back 16 bins (as the column pass would) into a checksum
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const cf t = z[zaddr((tid + 37 * q) & 63, (tid >> 2) & 63)];
    acc += t.x - t.y;
  }
  if (patch < n_dump)
    for (int i = tid; i < N * N; i += 256) dump[(size_t)patch * N * N + i] = make_float2(z[zaddr(i / N, i % N)].x, z[zaddr(i / N, i % N)].y);
  acc += __shfl_xor(acc, 32, 64);
  if (lane == 0) atomicAdd(&sums[patch & 1023], acc);
}

// ---- MFMA form ------------------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// column n of W (64 x 64): n < 33 -> cos(2 pi n x / 64); n >= 33 -> sin(2 pi (n - 32) x / 64), u = n - 32 = 1..31
__global__ void __launch_bounds__(256) rows_mfma(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                 const _Float16* __restrict__ w_hi, const _Float16* __restrict__ w_lo,
                                                 float* __restrict__ sums, float2* __restrict__ dump, int n_dump, int n) {
  // spectra tile: A = FFT(cur rows), B = FFT(prev rows), 33 bins each, as K1 would consume them: [image][row][u]
  __shared__ cf zs[2 * N * 40];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // B operand (the DFT matrix) in registers: [k step][n tile] x (hi, lo); b[j] = W[16 s + 8 h + j][32 t + r]
  half8 bh[4][2], bl[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bh[s][t][j] = w_hi[(16 * s + 8 * h + j) * N + 32 * t + r];
        bl[s][t][j] = w_lo[(16 * s + 8 * h + j) * N + 32 * t + r];
      }
  // persistent: the DFT matrix stays in registers, the workgroup walks the patch pairs; the pixels of the NEXT pair are
  // requested before the products of the current one are formed.
  // A operand: wave w takes rows 32 (w & 1) .. +31 of image (w >> 1); lane (r, h) needs px[row r][16 s + 8 h + j]
  const uint8_t* base = (wave >> 1 ? prev : cur) + (size_t)(32 * (wave & 1) + r) * N + 8 * h;
  uint32_t px[4][2], nx[4][2];
  int patch = blockIdx.x;
  if (patch < n) {
#pragma unroll
    for (int s = 0; s < 4; ++s) __builtin_memcpy(px[s], base + (size_t)patch * N * N + 16 * s, 8);
  }
  for (; patch < n; patch += gridDim.x) {
    const int next = patch + gridDim.x;
    if (next < n) {
#pragma unroll
      for (int s = 0; s < 4; ++s) __builtin_memcpy(nx[s], base + (size_t)next * N * N + 16 * s, 8);
    }
    // u8 -> f16, two pixels per v_perm_b32 + v_pk_add_f16: the half 0x6400 | b is 1024 + b exactly
    half8 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      typedef _Float16 half2 __attribute__((ext_vector_type(2)));
      const half2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, px[s][d], 0x04010400u);  // (b0, 0x64, b1, 0x64)
        const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, px[s][d], 0x04030402u);  // (b2, 0x64, b3, 0x64)
        const half2 p01 = __builtin_bit_cast(half2, lo) - k1024, p23 = __builtin_bit_cast(half2, hi) - k1024;
        a[s][4 * d + 0] = p01.x;
        a[s][4 * d + 1] = p01.y;
        a[s][4 * d + 2] = p23.x;
        a[s][4 * d + 3] = p23.y;
      }
    }
    float16v acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bh[s][t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bl[s][t], acc[t], 0, 0, 0);
      }
    }
    // D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5). Tile 0 = cos columns 0..31 (Re F[u]); tile 1 =
    // column 32 (cos, u = 32) + sine columns u = 1..31 (Im F[u] = -sum x sin): Re and Im of bin r sit in the same lane.
    cf* out = zs + (wave >> 1) * N * 40 + 32 * (wave & 1) * 40;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      out[row * 40 + r] = {acc[0][i], r == 0 ? 0.f : -acc[1][i]};
      if (r == 0) out[row * 40 + 32] = {acc[1][i], 0.f};
    }
    __syncthreads();
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const cf t = zs[((tid + 37 * q) & 127) * 40 + ((tid >> 3) % 33)];
      sacc += t.x - t.y;
    }
    if (patch < n_dump)
      for (int i = tid; i < 2 * N * 33; i += 256) dump[(size_t)patch * 2 * N * 33 + i] = make_float2(zs[(i / 33) * 40 + i % 33].x, zs[(i / 33) * 40 + i % 33].y);
    sacc += __shfl_xor(sacc, 32, 64);
    if (lane == 0) atomicAdd(&sums[patch & 1023], sacc);
    __syncthreads();  // the tile is rewritten by the next pair
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      px[s][0] = nx[s][0];
      px[s][1] = nx[s][1];
    }
  }
}

template <class F>
float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1, nullptr));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

}  // namespace

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 65536, reps = argc > 2 ? std::atoi(argv[2]) : 20, n_dump = 8;
  const size_t bytes = (size_t)n * N * N;
  std::vector<uint8_t> hc(bytes), hp(bytes);
  uint32_t st = 12345u;
  for (size_t i = 0; i < bytes; ++i) {
    st = st * 1664525u + 1013904223u;
    hc[i] = (uint8_t)(st >> 24);
    st = st * 1664525u + 1013904223u;
    hp[i] = (uint8_t)(st >> 24);
  }
  std::vector<float> tw(2 * N);
  std::vector<_Float16> whi((size_t)N * N), wlo((size_t)N * N);
  const double PI = 3.14159265358979323846;
  for (int k = 0; k < N; ++k) {
    tw[2 * k] = (float)std::cos(-2 * PI * k / N);
    tw[2 * k + 1] = (float)std::sin(-2 * PI * k / N);
  }
  for (int x = 0; x < N; ++x)
    for (int c = 0; c < N; ++c) {
      const double w = c <= 32 ? std::cos(2 * PI * c * x / N) : std::sin(2 * PI * (c - 32) * x / N);
      const _Float16 hi = (_Float16)w;
      whi[(size_t)x * N + c] = hi;
      wlo[(size_t)x * N + c] = (_Float16)(w - (double)hi);
    }
  uint8_t *dc, *dp;
  float *dtw, *dsums;
  _Float16 *dwh, *dwl;
  float2 *dump_v, *dump_m;
  CHECK(hipMalloc(&dc, bytes));
  CHECK(hipMalloc(&dp, bytes));
  CHECK(hipMalloc(&dtw, tw.size() * 4));
  CHECK(hipMalloc(&dsums, 1024 * 4));
  CHECK(hipMalloc(&dwh, whi.size() * 2));
  CHECK(hipMalloc(&dwl, wlo.size() * 2));
  CHECK(hipMalloc(&dump_v, (size_t)n_dump * N * N * 8));
  CHECK(hipMalloc(&dump_m, (size_t)n_dump * 2 * N * 33 * 8));
  CHECK(hipMemcpy(dc, hc.data(), bytes, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dp, hp.data(), bytes, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dtw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dwh, whi.data(), whi.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dwl, wlo.data(), wlo.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemset(dsums, 0, 4096));

  const float ms_v = time_ms([&] { hipLaunchKernelGGL(rows_valu, dim3(n), dim3(256), 0, nullptr, dc, dp, dtw, dsums, dump_v, n_dump); }, reps);
  const float ms_m = time_ms([&] { hipLaunchKernelGGL(rows_mfma, dim3(n < 768 ? n : 768), dim3(256), 0, nullptr, dc, dp, dwh, dwl, dsums, dump_m, n_dump, n); }, reps);
  CHECK(hipDeviceSynchronize());

  // validation against a double-precision DFT of the first patches
  std::vector<float2> hv((size_t)n_dump * N * N), hm((size_t)n_dump * 2 * N * 33);
  CHECK(hipMemcpy(hv.data(), dump_v, hv.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hm.data(), dump_m, hm.size() * 8, hipMemcpyDeviceToHost));
  double err_v = 0, err_m = 0, mag = 0;
  for (int pch = 0; pch < n_dump; ++pch)
    for (int row = 0; row < N; ++row)
      for (int u = 0; u < N; ++u) {
        double cr = 0, ci = 0, pr = 0, pi = 0;
        for (int x = 0; x < N; ++x) {
          const double a = -2 * PI * u * x / N, c = std::cos(a), s = std::sin(a);
          const double vc = hc[(size_t)pch * N * N + row * N + x], vp = hp[(size_t)pch * N * N + row * N + x];
          cr += vc * c; ci += vc * s; pr += vp * c; pi += vp * s;
        }
        // packed transform Z = FFT(cur) + i FFT(prev)
        const float2 gv = hv[(size_t)pch * N * N + row * N + u];
        err_v = std::fmax(err_v, std::fmax(std::fabs(gv.x - (cr - pi)), std::fabs(gv.y - (ci + pr))));
        mag = std::fmax(mag, std::hypot(cr, ci));
        if (u <= 32) {
          const float2 ga = hm[(size_t)pch * 2 * N * 33 + (size_t)row * 33 + u], gb = hm[(size_t)pch * 2 * N * 33 + (size_t)(N + row) * 33 + u];
          err_m = std::fmax(err_m, std::fmax(std::fmax(std::fabs(ga.x - cr), std::fabs(ga.y - ci)), std::fmax(std::fabs(gb.x - pr), std::fabs(gb.y - pi))));
        }
      }
  const double gb = 2.0 * bytes / 1e9;
  std::printf("{\"patch_pairs\": %d, \"valu_rows_ms\": %.4f, \"mfma_rows_ms\": %.4f, \"valu_GBps\": %.1f, \"mfma_GBps\": %.1f, "
              "\"max_abs_err_valu\": %.3e, \"max_abs_err_mfma\": %.3e, \"max_bin_magnitude\": %.1f, "
              "\"mfma_issued_per_pair\": 64, \"mfma_tflops_issued\": %.1f}\n",
              n, ms_v, ms_m, gb / (ms_v * 1e-3), gb / (ms_m * 1e-3), err_v, err_m, mag,
              (double)n * 64 * 32768 / (ms_m * 1e-3) / 1e12);
  return 0;
}
