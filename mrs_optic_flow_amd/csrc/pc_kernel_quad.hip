// pc_kernel_quad.hip -- K1 for 64 x 64 patches, "quad per line" formulation (gfx950 / CDNA4).
//
// EVALUATED ALTERNATIVE, off by default (MOF_PC_QUAD=1 selects it for N = 64; tests/test_gpu_fft.py keeps it correct).
// Same algorithm, same arithmetic and the same references as pc_kernel.hip; what changes is how the four 1-D passes
// move data. pc_kernel.hip runs each 64-point transform as two Stockham stages through LDS (two tile reads + two tile
// writes per pass). Here a 64-point line belongs to the
// four lanes of a QUAD: 64 = 4 x 16, sixteen points per lane in registers (radix-16 butterfly) and the radix-4 step
// ACROSS the lanes done with DPP quad_perm operands on v_fmac (no LDS, no extra instruction for the exchange):
//
//   A  load 2 x 16 B per lane, forward row transform in registers, ONE tile write
//   B  ONE tile read, forward column transform, ONE tile write
//   C  two half-tile reads (bin and Hermitian partner), normalised cross-power spectrum in registers, column
//      transform of the half spectrum (columns 0..31; column 0 carries columns 0 and 32, both real after the
//      transform), half-tile write
//   D  half-tile read, two output rows per complex transform, arg-max from registers; only the few rows of the
//      centroid window are written back
//
// 2.5 tile reads + 2.5 tile writes per patch instead of 8 + 6.5, every access pattern free of bank conflicts
// (tools/design/quad_fft_emulator.py replays the lane maps, sign tricks and bank maths against numpy.fft).
// Measured on MI355X (c2, 65,536 patches per launch): LDS instructions 210 -> 98 per wave, LDS busy 57 % -> 20 %, but
// VALU instructions 1335 -> 1805 per wave (the cross-lane radix-4 costs six VALU slots per complex value, work the
// Stockham form gets from the LDS for free) and 0.81 ms against 0.74 ms: the LDS pipe is not what binds K1 (DESIGN.md
// section 4: the latency of the per-patch phase chain at 4 workgroups per CU does), so its savings buy nothing and the
// extra VALU work costs ~9 %. Kept as a tested alternative and as the worked example of v_fmac_f32 with DPP operands.
//
// Lane maps (q = lane & 3, K1 = {0, 2, 1, 3}):
//   "blocked in"     lane q holds x[16 q + j]     -> exchange (xor 2, xor 1), twiddle, radix-16 -> X[K1[q] + 4 k]
//   "interleaved in" lane q holds x[K1[q] + 4 j]  -> radix-16, twiddle, exchange (xor 1, xor 2) -> X[k + 16 q]
// Both kinds use ONE per-lane twiddle set sigma_q W64^(j K1[q]), sigma = {+,-,-,-}: the signs that the
// "own + s * partner" form of the exchange leaves behind are folded into it.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"

namespace mof {

namespace {

constexpr int N = 64, H = 32, PITCH = 68, T = 256, WAVES = 4;
constexpr int TILE = 63 * PITCH + 63 + 4 * 3 + 1;  // complex elements
constexpr int WIN_ROWS = 7;                         // 5 (cv::phaseCorrelate) or 7 (OpenCL model) window rows
constexpr size_t LDS_BYTES_Q64 = sizeof(cf) * TILE + sizeof(Best) * WAVES + sizeof(float) * WIN_ROWS * N;

// skewed tile: rows 16 apart are shifted by 4 elements so that row-blocked column walks hit distinct banks
__device__ __forceinline__ int zq(int v, int u) { return v * PITCH + u + 4 * (v >> 4); }

// Value of lane quad_perm[q] of the same quad. MOF_QUAD_XCHG selects the carrier: 0 = DPP (a v_mov_b32_dpp on the
// VALU; hipcc does not fold it into the consuming v_fmac), 1 = ds_swizzle_b32 in quad-perm mode (the LDS crossbar, no
// memory access), 2 = v_fmac_f32 with a DPP operand, hand-written (see quad_radix4).
#ifndef MOF_QUAD_XCHG
#define MOF_QUAD_XCHG 2
#endif
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
#if MOF_QUAD_XCHG == 1
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x8000 | CTRL));
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
#endif
}
constexpr int QP_XOR1 = 0xB1;  // quad_perm [1,0,3,2]
constexpr int QP_XOR2 = 0x4E;  // quad_perm [2,3,0,1]
constexpr int QP_0132 = 0xB4;  // quad_perm [0,1,3,2]

// radix-4 across the quad: t = own + sa * partner(FIRST); lane 3 turns by +i; own + sb * partner(SECOND)
#if MOF_QUAD_XCHG == 2
// Hand-scheduled: v_fmac_f32 with a DPP operand does exchange and butterfly in ONE instruction (hipcc never folds the
// v_mov_b32_dpp into the fmac). The compiler cannot see DPP reads inside inline asm, so the block provides the wait
// states itself: s_nop 1 covers "VALU wrote the register a DPP operand reads" for whatever precedes the block, and
// inside it every DPP read is at least eight instructions after the write of that register.
#define MOF_QR4_BLOCK(FIRST, SECOND)                                                                                   \
  asm("s_nop 1\n\t"                                                                                                    \
      "v_fmac_f32_dpp %0, %0, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %4, %4, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %1, %1, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %5, %5, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %2, %2, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %6, %6, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %3, %3, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_fmac_f32_dpp %7, %7, %12 quad_perm:" FIRST " row_mask:0xf bank_mask:0xf\n\t"                                    \
      "v_cndmask_b32_e64 %8, %0, -%4, %14\n\t"                                                                          \
      "v_cndmask_b32_e64 %4, %4, %0, %14\n\t"                                                                           \
      "v_cndmask_b32_e64 %9, %1, -%5, %14\n\t"                                                                          \
      "v_cndmask_b32_e64 %5, %5, %1, %14\n\t"                                                                           \
      "v_cndmask_b32_e64 %10, %2, -%6, %14\n\t"                                                                         \
      "v_cndmask_b32_e64 %6, %6, %2, %14\n\t"                                                                           \
      "v_cndmask_b32_e64 %11, %3, -%7, %14\n\t"                                                                         \
      "v_cndmask_b32_e64 %7, %7, %3, %14\n\t"                                                                           \
      "v_fmac_f32_dpp %8, %8, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                   \
      "v_fmac_f32_dpp %4, %4, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                   \
      "v_fmac_f32_dpp %9, %9, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                   \
      "v_fmac_f32_dpp %5, %5, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                   \
      "v_fmac_f32_dpp %10, %10, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                 \
      "v_fmac_f32_dpp %6, %6, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                   \
      "v_fmac_f32_dpp %11, %11, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf\n\t"                                 \
      "v_fmac_f32_dpp %7, %7, %13 quad_perm:" SECOND " row_mask:0xf bank_mask:0xf"                                       \
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "=&v"(t0), "=&v"(t1),         \
        "=&v"(t2), "=&v"(t3)                                                                                           \
      : "v"(sa), "v"(sb), "s"(q3mask))

template <int FIRST, int SECOND>
__device__ __forceinline__ void quad_radix4(cf* v, float sa, float sb, bool /*q3*/) {
  const unsigned long long q3mask = 0x8888888888888888ull;  // lanes with (lane & 3) == 3
#pragma unroll
  for (int k = 0; k < 16; k += 4) {
    float x0 = v[k].x, x1 = v[k + 1].x, x2 = v[k + 2].x, x3 = v[k + 3].x;
    float y0 = v[k].y, y1 = v[k + 1].y, y2 = v[k + 2].y, y3 = v[k + 3].y;
    float t0, t1, t2, t3;
    if constexpr (FIRST == QP_XOR2) MOF_QR4_BLOCK("[2,3,0,1]", "[1,0,3,2]");
    else MOF_QR4_BLOCK("[1,0,3,2]", "[2,3,0,1]");
    v[k] = {t0, y0};
    v[k + 1] = {t1, y1};
    v[k + 2] = {t2, y2};
    v[k + 3] = {t3, y3};
  }
}
#else
template <int FIRST, int SECOND>
__device__ __forceinline__ void quad_radix4(cf* v, float sa, float sb, bool q3) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float tx = __builtin_fmaf(dpp<FIRST>(v[k].x), sa, v[k].x);
    const float ty = __builtin_fmaf(dpp<FIRST>(v[k].y), sa, v[k].y);
    const float rx = q3 ? -ty : tx, ry = q3 ? tx : ty;
    v[k].x = __builtin_fmaf(dpp<SECOND>(rx), sb, rx);
    v[k].y = __builtin_fmaf(dpp<SECOND>(ry), sb, ry);
  }
}
#endif

struct QuadConst {
  float dit_sa, dit_sb, dif_sa, dif_sb;
  bool q3;
  cf tw[16];
};

// lane q holds x[16 q + j] -> X[K1[q] + 4 k]
__device__ __forceinline__ void pass_blocked_in(cf* v, const QuadConst& c) {
  quad_radix4<QP_XOR2, QP_XOR1>(v, c.dit_sa, c.dit_sb, c.q3);
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = cmul(v[j], c.tw[j]);
  butterfly<16>(v);
}

// lane q holds x[K1[q] + 4 j] -> X[k + 16 q]
__device__ __forceinline__ void pass_interleaved_in(cf* v, const QuadConst& c) {
  butterfly<16>(v);
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], c.tw[k]);
  quad_radix4<QP_XOR1, QP_XOR2>(v, c.dif_sa, c.dif_sb, c.q3);
}

// 16 pixels (row, col..col+15) of one image of the patch as floats; see pc_kernel.hip for DS / CH
template <int DS, int CH>
__device__ __forceinline__ void load16(const uint8_t* base, size_t pitch, int row, int col, float* out) {
  if constexpr (DS == 1) {
    uint32_t w[4];
    if constexpr (CH == 1) __builtin_memcpy(w, base + (size_t)row * pitch + col, 16);
    else gray16_from_bgr48(base + (size_t)row * pitch + 3 * col, w);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = (float)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
  } else {
    // long-range mode: rounded mean of the 2x2 centre of each 4x4 cell (cv::resize 1/4, FftMethod.cpp:1931-1932)
    const uint8_t* r1 = base + (size_t)(4 * row + 1) * pitch + 4 * col;
#pragma unroll
    for (int h4 = 0; h4 < 4; ++h4) {
      uint32_t ra[4], rb[4];
      __builtin_memcpy(ra, r1 + 16 * h4, 16);
      __builtin_memcpy(rb, r1 + pitch + 16 * h4, 16);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t s = ((ra[b] >> 8) & 0xffu) + ((ra[b] >> 16) & 0xffu) + ((rb[b] >> 8) & 0xffu) + ((rb[b] >> 16) & 0xffu);
        out[4 * h4 + b] = (float)((s + 2u) >> 2);
      }
    }
  }
}

}  // namespace

template <int DS, int CH, int PK>
__global__ void __launch_bounds__(T, 4) pc_quad64_kernel(PcArgs a) {
  static_assert(CH == 1 || (CH == 3 && DS == 1), "BGR front end only for the full-resolution path");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  Best* red = reinterpret_cast<Best*>(z + TILE);
  float* win = reinterpret_cast<float*>(red + WAVES);

  const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
  // Every phase re-derives its lane coordinates from a laundered copy of the lane id: otherwise the address arithmetic
  // of all four phases is hoisted to the top of the kernel and kept live (30 VGPRs spilled at 4 waves / SIMD).
#define MOF_QUAD_LANE()                                  \
  int lane = lane0;                                      \
  asm volatile("" : "+v"(lane));                         \
  lane &= 63;                                            \
  const int g = lane >> 2, q = lane & 3;                 \
  const int k1 = (q == 1) ? 2 : (q == 2) ? 1 : q;        \
  (void)g; (void)k1

  const int patches = a.grid_x * a.grid_y;
  const int p = blockIdx.x;
  const int pr = p / patches, pt = p % patches;
  const int px0 = a.origin_x + (pt % a.grid_x) * a.stride_x, py0 = a.origin_y + (pt / a.grid_x) * a.stride_y;
  const size_t poff = (size_t)(DS * py0) * a.pitch + (size_t)(CH * DS * px0);
  const uint8_t* cur = a.cur + (size_t)pr * a.cur_stride + poff;
  const uint8_t* prev = a.prev + (size_t)pr * a.prev_stride + poff;

  QuadConst c;
  {
  MOF_QUAD_LANE();
  c.q3 = q == 3;
  c.dit_sa = q < 2 ? 1.f : -1.f;
  c.dit_sb = (q == 0 || q == 3) ? 1.f : -1.f;
  c.dif_sa = (q == 1 || q == 2) ? 1.f : -1.f;
  c.dif_sb = q >= 2 ? 1.f : -1.f;
  {
    const float sg = q ? -1.f : 1.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int idx = j * k1;  // < 64
      c.tw[j] = {sg * a.twiddles[2 * idx], sg * a.twiddles[2 * idx + 1]};
    }
  }
  }

  cf v[16];

  // ---- A: u8 -> f32 (exact), z = cur + i prev (convertTo :1805-1806), forward transform of the quad's row
  {
    MOF_QUAD_LANE();
    const int row = wave * 16 + g;
    float fc[16], fp[16];
#ifdef MOF_QABL_NOLOAD  // diagnostic build: no HBM access
#pragma unroll
    for (int j = 0; j < 16; ++j) { fc[j] = (float)((lane * 7 + j * 13 + p) & 255); fp[j] = (float)((lane * 11 + j * 5 + p) & 255); }
#else
    load16<DS, CH>(cur, a.pitch, row, 16 * q, fc);
    load16<DS, CH>(prev, a.pitch, row, 16 * q, fp);
#endif
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = {fc[j], fp[j]};
    pass_blocked_in(v, c);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[zq(row, k1 + 4 * k)] = v[k];
  }
  __syncthreads();
#if defined(MOF_QABL_STOP) && MOF_QABL_STOP == 1
  if (tid == 0) a.out[2 * (size_t)p] = z[p & 1023].x;
  return;
#endif

  // ---- B: forward transform of the quad's column (dft x2, :1491-1493)
  {
    MOF_QUAD_LANE();
    const int u = 4 * wave + (g & 3) + 16 * (g >> 2);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = lds_read(&z[zq(k1 + 4 * j, u)]);
    pass_interleaved_in(v, c);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[zq(k + 16 * q, u)] = v[k];
  }
  __syncthreads();
#if defined(MOF_QABL_STOP) && MOF_QABL_STOP == 2
  if (tid == 0) a.out[2 * (size_t)p] = z[p & 1023].x;
  return;
#endif

  // ---- C: untangle A = FFT(cur), B = FFT(prev); C = normalised cross-power (mulSpectrums :1494, magSpectrums :70-168,
  //      divSpectrums :1086-1251, real-only slots :107-109 / :1127-1129); column transform of D = conj(C) for columns
  //      0..31 (the other half is its mirror); column 0 carries D[.][0] + i D[.][32], whose transforms are both real.
  if (wave < 2) {
    MOF_QUAD_LANE();
    const int u = (g & 3) + 16 * ((g >> 2) & 1) + 4 * (g >> 3) + 8 * wave;
    const int um = (N - u) & (N - 1);
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {  // four bins at a time: bounds the LDS reads in flight (register pressure)
#pragma unroll
      for (int j = j0; j < j0 + 4; ++j) {
        const int vr = j + 16 * q, vm = (N - vr) & (N - 1);
        const cf zk = lds_read(&z[zq(vr, u)]), zm = lds_read(&z[zq(vm, um)]);
        const cf C = cross_power<PK>(zk, zm, false);
        v[j] = {C.x, -C.y};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pass_blocked_in(v, c);
    if (u != 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) z[zq(k1 + 4 * k, u)] = v[k];
    }
  } else if (wave == 3 && lane0 < 4) {
    // One quad of an otherwise idle wave takes the packed column D[.][0] + i D[.][32] (twice the cross-power work of
    // a regular column) in parallel: on wave 0 it would lengthen the critical path of the phase by 40 %.
    MOF_QUAD_LANE();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int vr = j + 16 * q, vm = (N - vr) & (N - 1);
      const bool self = j == 0 && (q == 0 || q == 2);  // rows 0 and 32: the four real-only CCS slots
      const cf C0 = cross_power<PK>(lds_read(&z[zq(vr, 0)]), lds_read(&z[zq(vm, 0)]), self);
      const cf Ch = cross_power<PK>(lds_read(&z[zq(vr, H)]), lds_read(&z[zq(vm, H)]), self);
      v[j] = {C0.x + Ch.y, Ch.x - C0.y};
    }
    pass_blocked_in(v, c);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[zq(k1 + 4 * k, 0)] = v[k];
  }
  __syncthreads();
#if defined(MOF_QABL_STOP) && MOF_QABL_STOP == 3
  if (tid == 0) a.out[2 * (size_t)p] = z[p & 1023].x;
  return;
#endif

  // ---- D: rows y and y + 32 ride one complex transform: E[u] = G[y][u] + i G[y+32][u], E[N-u] = conj(G[y][u]) +
  //      i conj(G[y+32][u]) (idft :1497, unscaled); arg-max from registers (fftShift :1297-1305, minMaxLoc :1539)
  //      Runs on waves 2 and 3 (phase C ran on waves 0 and 1): every SIMD gets the same share of a patch.
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  if (wave >= 2) {
    MOF_QUAD_LANE();
    const int y1 = 16 * (wave - 2) + g;  // un-shifted rows y1 and y1 + 32 -> shifted rows y1 + 32 and y1
    cf fm[9];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int u = k1 + 4 * j;
      const cf ga = lds_read(&z[zq(y1, u)]), gb = lds_read(&z[zq(y1 + H, u)]);
      cf e = {ga.x - gb.y, ga.y + gb.x};
      const cf f = {ga.x + gb.y, gb.x - ga.y};
      if (j == 0) {
        if (q == 0) e = {ga.x, gb.x};  // packed column: real parts are column 0, imaginary parts column 32
        fm[8] = {ga.y, gb.y};          // E[32] (lane 0 only)
      }
      v[j] = e;
      fm[j] = f;
    }
    // mirrored values travel to the lane that owns N - u: classes 1 <-> 3 sit on lanes 2 <-> 3; lane 0 owns 4 i and
    // 64 - 4 i itself, one slot later than the other lanes
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const cf m = {q == 0 ? fm[i + 1].x : fm[i].x, q == 0 ? fm[i + 1].y : fm[i].y};
      v[15 - i] = {dpp<QP_0132>(m.x), dpp<QP_0132>(m.y)};
    }
    pass_interleaved_in(v, c);
    // v[k].x = c[y1][k + 16 q], v[k].y = c[y1 + 32][k + 16 q]
    if constexpr (PK == 1) {  // OpenCL-kernel model: 1/N^2 scaling and the +-search_radius mask (cl:733, :737-746, :823-826)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        v[k].x = ocl_scale_mask<N>(v[k].x, y1, k + 16 * q, a.search_radius);
        v[k].y = ocl_scale_mask<N>(v[k].y, y1 + H, k + 16 * q, a.search_radius);
      }
    }
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < 16; ++k) m = fmaxf(m, fmaxf(v[k].x, v[k].y));
    int mi = 0x7fffffff;
    const int xs0 = 16 * ((q + 2) & 3);  // shifted column of k = 0
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      mi = min(mi, v[k].x == m ? (y1 + H) * N + xs0 + k : 0x7fffffff);
      mi = min(mi, v[k].y == m ? y1 * N + xs0 + k : 0x7fffffff);
    }
    best = Best{m, mi};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
      best = better(best, o);
    }
    if (lane == 0) red[wave] = best;
  }
  __syncthreads();
  best = better(red[2], red[3]);

  // ---- the rows of the centroid window go to LDS (at most 2 x RAD + 1 rows, from the lanes that hold them)
  constexpr int RAD = PeakModel<PK>::RAD;
  const int wy0 = best.idx / N - RAD;  // first window row (shifted coordinates; may be negative)
  if (wave >= 2) {
    MOF_QUAD_LANE();
    const int y1 = 16 * (wave - 2) + g;
    const int xs0 = 16 * ((q + 2) & 3);
    const int ra = y1 + H - wy0, rb = y1 - wy0;  // window row of the .x / .y values
    if (ra >= 0 && ra <= 2 * RAD) {
#pragma unroll
      for (int k = 0; k < 16; ++k) win[ra * N + xs0 + k] = v[k].x;
    }
    if (rb >= 0 && rb <= 2 * RAD) {
#pragma unroll
      for (int k = 0; k < 16; ++k) win[rb * N + xs0 + k] = v[k].y;
    }
  }
  __syncthreads();

  // ---- weighted centroid in double + validity gate (:1337-1383, :1838-1856), wave 0
  if (wave == 0) {
    const int lane = lane0;
    const float wval = centroid_window_value<N, PK>(best, lane, [&](int ys, int xs) { return win[(ys - wy0) * N + xs]; });
    centroid_gate_store<N, PK>(best, wval, lane, a.max_px_speed_sq, a.out + 2 * (size_t)p);
  }
}

#undef MOF_QUAD_LANE

template <int DS, int CH, int PK>
static hipError_t configure_quad_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_quad64_kernel<DS, CH, PK>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_Q64);
}

hipError_t pc_configure_quad64() {
  hipError_t e;
  if ((e = configure_quad_one<1, 1, 0>()) != hipSuccess) return e;
  if ((e = configure_quad_one<1, 3, 0>()) != hipSuccess) return e;
  if ((e = configure_quad_one<4, 1, 0>()) != hipSuccess) return e;
  if ((e = configure_quad_one<1, 1, 1>()) != hipSuccess) return e;
  if ((e = configure_quad_one<1, 3, 1>()) != hipSuccess) return e;
  return configure_quad_one<4, 1, 1>();
}

hipError_t launch_pc_field_quad64(const PcArgs& a_in, int n_pairs, hipStream_t stream) {
  PcArgs a = a_in;
  a.total = n_pairs * a.grid_x * a.grid_y;
  if (a.downscale == 4 && a.channels == 3) return hipErrorInvalidValue;
  const dim3 g((unsigned)a.total), b(T);
  if (a.peak_model == 1) {
    if (a.downscale == 4) hipLaunchKernelGGL((pc_quad64_kernel<4, 1, 1>), g, b, LDS_BYTES_Q64, stream, a);
    else if (a.channels == 3) hipLaunchKernelGGL((pc_quad64_kernel<1, 3, 1>), g, b, LDS_BYTES_Q64, stream, a);
    else hipLaunchKernelGGL((pc_quad64_kernel<1, 1, 1>), g, b, LDS_BYTES_Q64, stream, a);
  } else {
    if (a.downscale == 4) hipLaunchKernelGGL((pc_quad64_kernel<4, 1, 0>), g, b, LDS_BYTES_Q64, stream, a);
    else if (a.channels == 3) hipLaunchKernelGGL((pc_quad64_kernel<1, 3, 0>), g, b, LDS_BYTES_Q64, stream, a);
    else hipLaunchKernelGGL((pc_quad64_kernel<1, 1, 0>), g, b, LDS_BYTES_Q64, stream, a);
  }
  return hipGetLastError();
}

}  // namespace mof
