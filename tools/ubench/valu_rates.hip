// Micro-benchmark: sustained issue rate of the VALU instructions the kernels depend on (gfx950).
// Each kernel runs a long unrolled stream of one instruction on 8 independent accumulators per lane,
// 8 waves per SIMD on every CU; reports cycles per wave-instruction per SIMD (at the measured clock).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define ITERS 2048
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 * 3u;
  uint64_t q[8];
  float f[8], g[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) { q[i] = a0 + i; f[i] = (float)(a1 & 255) + i; g[i] = 1.0f + i * 1e-3f; p[i] = f2{f[i], g[i]}; u[i] = a2 + i; }
  const uint64_t w = ((uint64_t)a3 << 32) | a2;
  const f2 pc = {1.0001f, 0.9999f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(w, a1, q[i]);
      if (OP == 1) u[i] = __builtin_amdgcn_sad_u8(a2, a1, u[i]);
      if (OP == 2) f[i] = __builtin_fmaf(f[i], g[i], 1.5f);
      if (OP == 3) f[i] = f[i] + g[i];
      if (OP == 4) p[i] = p[i] + pc;
      if (OP == 5) p[i] = __builtin_elementwise_fma(p[i], pc, pc);
      if (OP == 6) u[i] = __builtin_amdgcn_sad_u16(a2, a1, u[i]);
      if (OP == 7) u[i] = __builtin_amdgcn_msad_u8(a2, a1, u[i]);
      if (OP == 8) q[i] = __builtin_amdgcn_mqsad_pk_u16_u8(w, a1, q[i]);
      if (OP == 9) u[i] = u[i] * a1 + a2;                    // v_mad_u32_u24 / v_mul_lo
      if (OP == 10) u[i] = (u[i] & a1) | a2;                 // v_and_or
      if (OP == 11) f[i] = __builtin_amdgcn_rcpf(f[i]);
      if (OP == 12) u[i] = __builtin_amdgcn_alignbyte(u[i], a1, 1);
      if (OP == 13) u[i] = __builtin_amdgcn_update_dpp(0, u[i], 0x4E, 0xf, 0xf, true) + a1;   // v_mov_b32_dpp quad_perm + v_add
      if (OP == 14) f[i] = (u[i] & 1) ? f[i] : g[i] + f[i];                                     // v_add + v_cndmask
      if (OP == 15) f[i] = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f[i]), 0xB1, 0xf, 0xf, true)), g[i], f[i]);  // mov_dpp + fmac
      if (OP == 16) u[i] = __builtin_amdgcn_ds_swizzle(u[i], 0x8000 | 0x4E) + a1;              // ds_swizzle + v_add
      if (OP == 17) u[i] = __builtin_amdgcn_update_dpp(0, u[i], 0x128, 0xf, 0xf, true) + a1;  // row_ror:8
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ __float_as_uint(f[i]) ^ __float_as_uint(p[i].x) ^ __float_as_uint(p[i].y) ^ u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, uint32_t* d, int cus, double clock_ghz) {
  const int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double insts_per_simd = (double)ITERS * 8 * 8;  // 8 waves per SIMD, 8 instr per iteration
  const double cyc = ms * 1e-3 * clock_ghz * 1e9 / insts_per_simd;
  printf("%-22s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms, cyc, clock_ghz);
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("%s, %d CUs, clockRate %.2f GHz\n", prop.name, cus, ghz);
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
  run<2>("v_fma_f32", d, cus, ghz);
  run<3>("v_add_f32", d, cus, ghz);
  run<4>("v_pk_add_f32", d, cus, ghz);
  run<5>("v_pk_fma_f32", d, cus, ghz);
  run<0>("v_qsad_pk_u16_u8", d, cus, ghz);
  run<8>("v_mqsad_pk_u16_u8", d, cus, ghz);
  run<1>("v_sad_u8", d, cus, ghz);
  run<7>("v_msad_u8", d, cus, ghz);
  run<6>("v_sad_u16", d, cus, ghz);
  run<9>("u32 mul-add", d, cus, ghz);
  run<10>("v_and_or_b32", d, cus, ghz);
  run<11>("v_rcp_f32", d, cus, ghz);
  run<12>("v_alignbyte_b32", d, cus, ghz);
  run<13>("mov_dpp quad + add (2)", d, cus, ghz);
  run<14>("add + cndmask (2)", d, cus, ghz);
  run<15>("mov_dpp + fmac (2)", d, cus, ghz);
  run<16>("ds_swizzle + add (1+1)", d, cus, ghz);
  run<17>("mov_dpp row_ror + add (2)", d, cus, ghz);
  return 0;
}
