// pc_common.hpp -- device helpers shared by the phase-correlation kernels (pc_kernel.hip: power-of-two patch
// sizes; pc_kernel_mixed.hip: 120 = 15 x 8; sr_kernel.hip: whole frames of 240 / 256 / 480): complex arithmetic, radix butterflies, wave-local LDS ordering, the
// normalised cross-power spectrum of one bin and the centroid / gate tail. Reference citations as in pc_kernel.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mof_kernels.h"

namespace mof {
namespace {

struct cf {
  float x, y;
};

__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
// multiply by -i (a quarter turn of the forward kernel e^{-2 pi i k/N})
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }

template <int R>
__device__ __forceinline__ void butterfly(cf* v);

template <>
__device__ __forceinline__ void butterfly<2>(cf* v) {
  cf a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void butterfly<4>(cf* v) {
  cf a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
  cf c = cadd(v[1], v[3]), d = mul_mi(csub(v[1], v[3]));
  v[0] = cadd(a, c);
  v[1] = cadd(b, d);
  v[2] = csub(a, c);
  v[3] = csub(b, d);
}

template <>
__device__ __forceinline__ void butterfly<8>(cf* v) {
#ifdef MOF_ABLATE_NOBFLY
  return;  // diagnostic build: keeps the LDS traffic, drops the butterfly arithmetic
#endif
  const float h = 0.70710678118654752440f;
  cf e[4] = {v[0], v[2], v[4], v[6]};
  cf o[4] = {v[1], v[3], v[5], v[7]};
  butterfly<4>(e);
  butterfly<4>(o);
  cf w1 = {h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)};   // o1 * e^{-i pi/4}
  cf w2 = mul_mi(o[2]);                                     // o2 * e^{-i pi/2}
  cf w3 = {h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)};  // o3 * e^{-3 i pi/4}
  v[0] = cadd(e[0], o[0]);
  v[4] = csub(e[0], o[0]);
  v[1] = cadd(e[1], w1);
  v[5] = csub(e[1], w1);
  v[2] = cadd(e[2], w2);
  v[6] = csub(e[2], w2);
  v[3] = cadd(e[3], w3);
  v[7] = csub(e[3], w3);
}

// butterfly<8> of v_k = a_k t_k (t_0 = 1; tw[k-1] = t_k): the twiddle products ride the first radix-2 layer as FMAs
// (x + t y needs 4 FMAs, x - t y = 2 x - (x + t y) two more) -- 8 instructions fewer than 7 complex multiplies followed
// by butterfly<8>, the same value up to rounding.
__device__ __forceinline__ void butterfly8_tw(cf* v, const cf* tw) {
  auto fma_c = [](cf x, cf t, cf y) -> cf {  // x + t * y
    return {__builtin_fmaf(t.x, y.x, __builtin_fmaf(-t.y, y.y, x.x)), __builtin_fmaf(t.x, y.y, __builtin_fmaf(t.y, y.x, x.y))};
  };
  auto twice_minus = [](cf x, cf s) -> cf { return {__builtin_fmaf(2.f, x.x, -s.x), __builtin_fmaf(2.f, x.y, -s.y)}; };
  const float h = 0.70710678118654752440f;
  // even half: e = (a0, t2 a2, t4 a4, t6 a6)
  const cf A = fma_c(v[0], tw[3], v[4]), B = twice_minus(v[0], A);
  const cf u2 = cmul(v[2], tw[1]);
  const cf C = fma_c(u2, tw[5], v[6]), Dm = twice_minus(u2, C);
  const cf D = mul_mi(Dm);
  const cf e0 = cadd(A, C), e1 = cadd(B, D), e2 = csub(A, C), e3 = csub(B, D);
  // odd half: o = (t1 a1, t3 a3, t5 a5, t7 a7)
  const cf u1 = cmul(v[1], tw[0]);
  const cf A1 = fma_c(u1, tw[4], v[5]), B1 = twice_minus(u1, A1);
  const cf u3 = cmul(v[3], tw[2]);
  const cf C1 = fma_c(u3, tw[6], v[7]), Dm1 = twice_minus(u3, C1);
  const cf D1 = mul_mi(Dm1);
  const cf o0 = cadd(A1, C1), o1 = cadd(B1, D1), o2 = csub(A1, C1), o3 = csub(B1, D1);
  const cf w1 = {h * (o1.x + o1.y), h * (o1.y - o1.x)};
  const cf w2 = mul_mi(o2);
  const cf w3 = {h * (o3.y - o3.x), -h * (o3.x + o3.y)};
  v[0] = cadd(e0, o0);
  v[4] = csub(e0, o0);
  v[1] = cadd(e1, w1);
  v[5] = csub(e1, w1);
  v[2] = cadd(e2, w2);
  v[6] = csub(e2, w2);
  v[3] = cadd(e3, w3);
  v[7] = csub(e3, w3);
}

// A/B (-DMOF_BFLY_PRIO_IN=a -DMOF_BFLY_PRIO_OUT=b): issue priority of a wave inside its radix-16 butterflies against the LDS phases around
// them (r06, profiles/r06_bfly_prio_ab.txt: butterflies first -4 .. -5 % at c2 and -7 % at c4, LDS phases first +-0 -- off)
#if defined(MOF_BFLY_PRIO_IN)
#define MOF_BFLY_ENTER() __builtin_amdgcn_s_setprio(MOF_BFLY_PRIO_IN)
#define MOF_BFLY_LEAVE() __builtin_amdgcn_s_setprio(MOF_BFLY_PRIO_OUT)
#else
#define MOF_BFLY_ENTER() ((void)0)
#define MOF_BFLY_LEAVE() ((void)0)
#endif
template <>
__device__ __forceinline__ void butterfly<16>(cf* v) {
  MOF_BFLY_ENTER();
  // 16 = 4 x 4: four radix-4 over n1 (stride 4), twiddle W16^{n2 k1}, four radix-4 over n2
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  cf t[4][4];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    cf a[4] = {v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]};
    butterfly<4>(a);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) t[n2][k1] = a[k1];
  }
  // W16^m = (cos(pi m/8), -sin(pi m/8))
  const cf w[10] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}, {-1.f, 0.f},
                    {-c1, s1}};
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    cf a[4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) a[n2] = (n2 * k1 == 0) ? t[n2][k1] : cmul(t[n2][k1], w[n2 * k1]);
    butterfly<4>(a);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = a[k2];
  }
  MOF_BFLY_LEAVE();
}

__device__ __forceinline__ void butterfly3(cf* a) {
  const float s3 = 0.86602540378443864676f;  // sin(2 pi / 3)
  const cf s = cadd(a[1], a[2]), d = csub(a[1], a[2]);
  const cf m = {a[0].x - 0.5f * s.x, a[0].y - 0.5f * s.y};
  a[0] = cadd(a[0], s);
  a[1] = {m.x + s3 * d.y, m.y - s3 * d.x};
  a[2] = {m.x - s3 * d.y, m.y + s3 * d.x};
}

__device__ __forceinline__ void butterfly5(cf* a) {
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;  // cos(2pi/5), cos(4pi/5)
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;   // sin(2pi/5), sin(4pi/5)
  const cf s14 = cadd(a[1], a[4]), d14 = csub(a[1], a[4]), s23 = cadd(a[2], a[3]), d23 = csub(a[2], a[3]);
  const cf p1 = {a[0].x + c1 * s14.x + c2 * s23.x, a[0].y + c1 * s14.y + c2 * s23.y};
  const cf p2 = {a[0].x + c2 * s14.x + c1 * s23.x, a[0].y + c2 * s14.y + c1 * s23.y};
  const cf q1 = {s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y};
  const cf q2 = {s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y};
  a[0] = {a[0].x + s14.x + s23.x, a[0].y + s14.y + s23.y};
  a[1] = {p1.x + q1.y, p1.y - q1.x};  // p1 - i q1
  a[4] = {p1.x - q1.y, p1.y + q1.x};  // p1 + i q1
  a[2] = {p2.x + q2.y, p2.y - q2.x};
  a[3] = {p2.x - q2.y, p2.y + q2.x};
}

// 15-point DFT: n = 5 n1 + n2, k = k1 + 3 k2 (radix 3, twiddle W15^{n2 k1}, radix 5)
__device__ __forceinline__ void butterfly15(cf* v) {
  // W15^m = (cos(2 pi m / 15), -sin(2 pi m / 15)), m = 0..8
  const cf w15[9] = {{1.f, 0.f},
                     {0.91354545764260089550f, -0.40673664307580020775f},
                     {0.66913060635885821383f, -0.74314482547739423501f},
                     {0.30901699437494742410f, -0.95105651629515357212f},
                     {-0.10452846326765347140f, -0.99452189536827333692f},
                     {-0.5f, -0.86602540378443864676f},
                     {-0.80901699437494742410f, -0.58778525229247312917f},
                     {-0.97814760073380563793f, -0.20791169081775933710f},
                     {-0.97814760073380563793f, 0.20791169081775933710f}};
  cf t[5][3];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    cf a[3] = {v[n2], v[5 + n2], v[10 + n2]};
    butterfly3(a);
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) t[n2][k1] = (n2 * k1 == 0) ? a[k1] : cmul(a[k1], w15[n2 * k1]);
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    cf b[5] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1], t[4][k1]};
    butterfly5(b);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[k1 + 3 * k2] = b[k2];
  }
}

// 32-point DFT: n = 8 n1 + n2, k = k1 + 4 k2 (eight radix-4 over n1, twiddle W32^{n2 k1}, four radix-8 over n2)
__device__ __forceinline__ void butterfly32(cf* v) {
  // W32^m = (cos(pi m / 16), -sin(pi m / 16)), m = 0..21
  const float c1 = 0.98078528040323044913f, s1 = 0.19509032201612826785f;  // pi/16
  const float c2 = 0.92387953251128675613f, s2 = 0.38268343236508977173f;  // pi/8
  const float c3 = 0.83146961230254523708f, s3 = 0.55557023301960222474f;  // 3 pi/16
  const float h = 0.70710678118654752440f;
  const cf w[22] = {{1.f, 0.f}, {c1, -s1}, {c2, -s2}, {c3, -s3}, {h, -h}, {s3, -c3}, {s2, -c2}, {s1, -c1}, {0.f, -1.f},
                    {-s1, -c1}, {-s2, -c2}, {-s3, -c3}, {-h, -h}, {-c3, -s3}, {-c2, -s2}, {-c1, -s1}, {-1.f, 0.f},
                    {-c1, s1}, {-c2, s2}, {-c3, s3}, {-h, h}, {-s3, c3}};
  cf t[8][4];
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) {
    cf a[4] = {v[n2], v[8 + n2], v[16 + n2], v[24 + n2]};
    butterfly<4>(a);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) t[n2][k1] = (n2 * k1 == 0) ? a[k1] : cmul(a[k1], w[n2 * k1]);
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    cf b[8] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1], t[4][k1], t[5][k1], t[6][k1], t[7][k1]};
    butterfly<8>(b);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k1 + 4 * k2] = b[k2];
  }
}

// Orders the LDS traffic of the lanes of ONE wave (a wave's DS instructions execute in order);
// emits no instruction, only stops the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8-byte LDS read kept as ONE ds_read_b64: hipcc otherwise fuses neighbouring reads into ds_read2_b64 /
// ds_read2st64_b64, which move half the bytes per LDS cycle on gfx950 (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ cf lds_read(const cf* p) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef const volatile f2 __attribute__((address_space(3))) * lds_f2_ptr;
  const f2 t = *(lds_f2_ptr)(p);
  return {t.x, t.y};
}

struct Best {
  float v;
  int idx;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  // first maximum in row-major order of the fft-shifted surface (cv::minMaxLoc)
  return (b.v > a.v || (b.v == a.v && b.idx < a.idx)) ? b : a;
}


// cv::cvtColor(.., CV_RGB2GRAY) on 8-bit data (fixed point, yuv_shift 14): gray = (c0*R2Y + c1*G2Y + c2*B2Y + 2^13) >> 14
// with R2Y 4899, G2Y 9617, B2Y 1868. The node feeds it BGR8 data (optic_flow.cpp:1465 toCvCopy(BGR8), :1622
// CV_RGB2GRAY), so the blue byte gets the red weight -- reproduced as is.
__device__ __forceinline__ uint32_t rgb2gray_fixed(uint32_t c0, uint32_t c1, uint32_t c2) {
  return (c0 * 4899u + c1 * 9617u + c2 * 1868u + 8192u) >> 14;
}

// 16 gray pixels from 48 interleaved bytes (12 dwords, any alignment)
__device__ __forceinline__ void gray16_from_bgr48(const uint8_t* p, uint32_t* g /*[4] packed u8x4*/) {
  uint32_t w[12];
  __builtin_memcpy(w, p, 48);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t packed = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = 3 * (4 * q + b);  // byte index of pixel 4q+b
      const uint32_t c0 = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
      const uint32_t c1 = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xffu;
      const uint32_t c2 = (w[(i + 2) >> 2] >> (8 * ((i + 2) & 3))) & 0xffu;
      packed |= rgb2gray_fixed(c0, c1, c2) << (8 * b);
    }
    g[q] = packed;
  }
}

// 8 gray pixels from 24 interleaved bytes (6 dwords, any alignment)
__device__ __forceinline__ void gray8_from_bgr24(const uint8_t* p, uint32_t* g /*[2] packed u8x4*/) {
  uint32_t w[6];
  __builtin_memcpy(w, p, 24);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    uint32_t packed = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = 3 * (4 * q + b);
      const uint32_t c0 = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
      const uint32_t c1 = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xffu;
      const uint32_t c2 = (w[(i + 2) >> 2] >> (8 * ((i + 2) & 3))) & 0xffu;
      packed |= rgb2gray_fixed(c0, c1, c2) << (8 * b);
    }
    g[q] = packed;
  }
}

// One bin of the normalised cross-power spectrum from the packed transform Z = FFT(cur + i prev):
//   A[k] = (Z[k] + conj(Z[-k]))/2 ; B[k] = (Z[k] - conj(Z[-k]))/(2i) ; P = A conj(B)        (mulSpectrums :1494)
//   C = P |P| / (|P|^2 + eps)                                   (magSpectrums :70-168, divSpectrums :1086-1251)
//   real-only CCS slots: C = P / (P^2 + eps)                                   (:107-109 / :1127-1129)
// ONE hardware sqrt and ONE hardware reciprocal (1 ulp each): the IEEE divide/sqrt expansions were a fifth of the
// kernel's VALU work for no effect at 1e-4 px.
//
// PK = 1 is the useOCL=true normalisation (cl/FftMethod.cl:971-982, :1024-1031): C = P rsqrt(|P|^2 + eps)
// for every pair and 1 / (a b) in the four real-only slots.
// cross_power_ab takes the two spectra already separated (and doubled): A2 = 2A, B2 = 2B.
template <int PK = 0>
__device__ __forceinline__ cf cross_power_ab(cf A, cf B, bool real_only) {
  // Worked on 2A = Z[k] + conj(Z[-k]) and 2B = -i (Z[k] - conj(Z[-k])): P' = 2A conj(2B) = 4P, and
  //   P |P| / (|P|^2 + eps)  ==  P' |P'| / (|P'|^2 + 16 eps)   exactly (powers of two), four multiplies fewer per bin.
  const float eps16 = 16.f * 1.1920928955078125e-07f;  // 16 * FLT_EPSILON, :1117
  if (real_only) {
    // P = A.x B.x / 4 ; C = P / (P^2 + eps) = 4 P' / (P'^2 + 16 eps)
    const float p4 = A.x * B.x;
    if constexpr (PK == 1) return {4.f * __builtin_amdgcn_rcpf(p4), 0.f};  // 1 / (a b), cl:1029
    return {4.f * p4 * __builtin_amdgcn_rcpf(p4 * p4 + eps16), 0.f};
  }
  const float pr = A.x * B.x + A.y * B.y;
  const float pim = A.y * B.x - A.x * B.y;
  const float q = pr * pr + pim * pim;
  if constexpr (PK == 1) {
    // P rsqrt(|P|^2 + eps) == P' rsqrt(|P'|^2 + 16 eps)
    const float s = __builtin_amdgcn_rsqf(q + eps16);
    return {pr * s, pim * s};
  }
  // |P'|^2 >= 2^10: 16 eps / q < 2^-30 is below half an ulp of the unit-magnitude result, so C = P' rsq(q) (one
  // transcendental); the general form only runs for (numerically) empty bins such as constant patches
  // -- taken as a wave-uniform branch: a per-lane one costs exec-mask juggling around every bin and, in the quad
  // kernel, 48 spilled VGPRs.
  float s = __builtin_amdgcn_rsqf(q);
  if (__builtin_amdgcn_ballot_w64(!(q >= 1024.f)) != 0)
    s = (q >= 1024.f) ? s : __builtin_amdgcn_sqrtf(q) * __builtin_amdgcn_rcpf(q + eps16);
  return {pr * s, pim * s};
}

// The two real images ride one complex transform (z = cur + i prev): untangle bin k from Z[k] and Z[-k], doubled
__device__ __forceinline__ void untangle2(cf zk, cf zm, cf* A2, cf* B2) {
  *A2 = {zk.x + zm.x, zk.y - zm.y};
  *B2 = {zk.y + zm.y, zm.x - zk.x};
}

template <int PK = 0>
__device__ __forceinline__ cf cross_power(cf zk, cf zm, bool real_only) {
  cf A, B;
  untangle2(zk, zm, &A, &B);
  return cross_power_ab<PK>(A, B, real_only);
}

// ---- degenerate pairs: a CONSTANT patch ----------------------------------------------------------------------------
// cv::phaseCorrelate transforms the two patches separately (FftMethod.cpp:1491-1493): a constant patch has an EXACTLY zero
// AC spectrum, the cross-power spectrum keeps its DC bin (a real-only slot: C = P / (P^2 + eps)) and nothing else, and the
// surface is flat = C_dc: first maximum at shifted (0, 0), centroid of the clamped 3 x 3 window of equal values =
// (1, 1) 9c / (9c + DBL_EPSILON) -- or (0, 0) when C_dc = 0 (an all-zero patch). The pair kernels pack cur + i prev into one
// complex transform, which leaks ~1e-7 of the textured patch's spectrum into those zeros; the normalisation blows that up
// to unit magnitude and the result is noise (found by tools/fft_sr_fuzz.py, r03). So constant patches are detected where
// the pixels are loaded (exact: integer compares) and the exact answer is substituted in the tail.
// A constant n x n patch that cv::phaseCorrelate pads to m x m is a BOX, and its spectrum -- level D[v] D[u], D the transform of n ones in a
// line of m -- is EXACTLY zero on every line k != 0 with k n = 0 (mod m), i.e. on the multiples of q = m / gcd(n, m): the Nyquist line
// alone when n and m share one factor of two only (158 in 160), but 40, 80, 120 for 156 in 160 and every multiple of 20 for 152 in 160.
// There P = 0 and C = 0; any transform that leaves rounding noise in those bins has it normalised to unit magnitude (0.03 - 0.09 px off on
// 156 / 152-pixel constant-against-texture pairs, tools/fft_sr_fuzz.py seeds 606 / 608, r05). The kernels know a constant patch exactly
// (integer compares where the pixels are loaded), so they zero exactly those bins. q = m when there is no such line.
__device__ __forceinline__ int box_zero_period(int n, int m) {
  int a = m, b = n;
  while (b != 0) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return m / a;
}
__device__ __forceinline__ bool box_zero_line(int k, int q) { return k != 0 && k % q == 0; }

// (1) per lane: `first` = its first pixel, `diff` != 0 iff one of its packed pixels differs from it
__device__ __forceinline__ void const_track(const uint32_t* words, int n_words, bool reset, uint32_t& first, uint32_t& diff) {
  if (reset) {
    first = words[0] & 0xffu;
    diff = 0u;
  }
  const uint32_t pat = first * 0x01010101u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < n_words) diff |= words[i] ^ pat;
}
// (2) per wave: the pixel value if every `on` lane holds nothing else, or 256 (lane 0 must be `on`)
__device__ __forceinline__ int wave_const_code(bool on, uint32_t first, uint32_t diff) {
  const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
  const bool bad = on && (diff != 0u || first != f0);
  return __builtin_amdgcn_ballot_w64(bad) == 0ull ? (int)f0 : 256;
}
// (3) the per-wave codes live in LDS as int codes[16]: cur code | prev code << 16, 256 = "not constant", written by lane 0
// of every wave for every patch (a textured patch fails a two-compare pre-test and writes 256 | 256 << 16); one wave reads
// them in the tail (the LDS read `my = codes[lane & 15]` is issued ahead of its other tail reads: the latencies overlap)
template <int WAVES>
__device__ __forceinline__ bool const_codes_degenerate(int my, int lane) {
  static_assert(WAVES <= 16, "16 codes");
  const int c = my & 0xffff, p = (int)((unsigned)my >> 16);
  const int c0 = __builtin_amdgcn_readfirstlane(c), p0 = __builtin_amdgcn_readfirstlane(p);
  const bool cur_const = c0 < 256 && __builtin_amdgcn_ballot_w64(lane < WAVES && c != c0) == 0ull;
  const bool prev_const = p0 < 256 && __builtin_amdgcn_ballot_w64(lane < WAVES && p != p0) == 0ull;
  return cur_const || prev_const;
}

// Weighted centroid in double + validity gate, executed by ONE wave in two steps so that the tile can be recycled in
// between: (1) (2 RAD + 1)^2 lanes fetch one window element each (`surface(ys, xs)` returns the fft-shifted correlation
// value; 0 outside the clamped window), (2) three fp64 sums are reduced by shuffles and lane 0 stores (x, y) or
// (NaN, NaN).
//   PK = 0: cv::phaseCorrelate's weightedCentroid, 5x5, every value, sum + DBL_EPSILON      (:1337-1383, :1838-1856)
//   PK = 1: the OpenCL kernel's refine(), 7x7, values > 0 only, sum seeded with FLT_EPSILON     (cl:1315-1379, :1478)
//           (the kernel sums floats over absolute frame coordinates; here patch-local doubles -- same value, without
//           the ~1e-4 px of rounding noise that representation adds)
template <int PK>
struct PeakModel {
  static constexpr int RAD = PK == 1 ? 3 : 2;
  static constexpr int W = 2 * RAD + 1;
};

template <int N, int PK = 0, class Surface>
__device__ __forceinline__ float centroid_window_value(Best best, int lane, Surface surface) {
  constexpr int RAD = PeakModel<PK>::RAD, W = PeakModel<PK>::W;
  const int px = best.idx % N, py = best.idx / N;
  const int ys = py - RAD + lane / W, xs = px - RAD + lane % W;
  if (lane < W * W && ys >= 0 && ys <= N - 1 && xs >= 0 && xs <= N - 1) {  // window clamped to the patch
    const float v = surface(ys, xs);
    if constexpr (PK == 1) return v > 0.f ? v : 0.f;
    return v;
  }
  return 0.f;
}

// degenerate: one of the two patches is constant; c_dc = the DC bin of the cross-power spectrum (see above)
template <int N, int PK = 0>
__device__ __forceinline__ void centroid_gate_store(Best best, float wval, int lane, double max_px_speed_sq, double* out,
                                                    bool degenerate = false, float c_dc = 0.f) {
  constexpr int RAD = PeakModel<PK>::RAD, W = PeakModel<PK>::W;
  const int px = best.idx % N, py = best.idx / N;
  const int ys = py - RAD + lane / W, xs = px - RAD + lane % W;
  const double val = (double)wval;  // 0 for lanes outside the window
  double cx = (double)xs * val, cy = (double)ys * val, sum = val;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cx += __shfl_xor(cx, off, 64);
    cy += __shfl_xor(cy, off, 64);
    sum += __shfl_xor(sum, off, 64);
  }
  if (lane == 0) {
    sum += PK == 1 ? 1.1920928955078125e-07 : 2.220446049250313e-16;  // FLT_EPSILON cl:1342 / DBL_EPSILON :1378
    // shift = -(center - t) = t - N/2   (:1836); the OpenCL branch returns centroid - N/2 un-negated (cl:1370, :1833)
    double sx = cx / sum - (double)N / 2.0;
    double sy = cy / sum - (double)N / 2.0;
    if (degenerate) {
      if constexpr (PK == 1) {
        sx = sy = __builtin_nan("");  // 1 / (a b) with b = 0 in the three other real-only slots (cl:1029): no finite surface
      } else {
        const double c9 = 9.0 * (double)c_dc;
        sx = sy = (c9 > 0.0 ? c9 / (c9 + 2.220446049250313e-16) : 0.0) - (double)N / 2.0;
      }
    }
    // best.idx == 0x7fffffff: no value compared equal to the maximum, i.e. the whole surface is NaN (a patch whose
    // spectrum holds infinities, e.g. 1/0 in a real-only slot under the OpenCL model) -> invalid, as a NaN centroid is
    const bool bad = (sx * sx + sy * sy > max_px_speed_sq) || (fabs(sx) > (double)N / 2.0) ||
                     (fabs(sy) > (double)N / 2.0) || (sx != sx) || (sy != sy) || (best.idx == 0x7fffffff && !degenerate);
    if (bad) sx = sy = __builtin_nan("");
    out[0] = sx;
    out[1] = sy;
  }
}

// PK = 1 only: value (y, x) of the UN-shifted surface after the kernel's scaling and +-search_radius mask
// (cl:733, :737-746, :823-826): rows / columns with search_radius < index < N - search_radius read as 0.
template <int N>
__device__ __forceinline__ float ocl_scale_mask(float v, int y, int x, int sr) {
  const bool masked = (y > sr && y < N - sr) || (x > sr && x < N - sr);
  return masked ? 0.f : v * (1.0f / (float)(N * N));
}

}  // namespace
}  // namespace mof
