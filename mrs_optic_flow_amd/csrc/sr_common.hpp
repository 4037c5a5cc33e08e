// sr_common.hpp -- the in-register two-stage 1-D transform shared by the scale/rotation kernels (sr_kernel.hip: independent
// pairs; sr_seq_kernel.hip: video sequences). Device code only.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "pc_common.hpp"
#include "pc_plan.hpp"  // butterfly_ct, butterfly10: the composite radices of the compile-time plans

namespace mof {
namespace {

template <int N>
struct SrPlan;
// LINE = complex elements per line buffer: >= R1 * Y2 (the padded stage-1 output), odd so that the 8 lines of a
// workgroup start on different banks; Y2 = row pitch of the stage-1 output (R2 + 1: stride-Y2 reads hit distinct banks)
#ifndef MOF_SR_LINE480  // (A/B: tools/ab_sr_line.sh)
#define MOF_SR_LINE480 497
#endif
// (A/B, r06: -DMOF_SR_480_R1=24 -DMOF_SR_480_R2=20 -DMOF_SR_LINE480=505 and the like -- the first radix may be up to 32 now)
#ifndef MOF_SR_480_R1
#define MOF_SR_480_R1 15
#define MOF_SR_480_R2 32
#endif
template <>
struct SrPlan<480> {
  static constexpr int R1 = MOF_SR_480_R1, R2 = MOF_SR_480_R2, Y2 = R2 + 1, LINE = MOF_SR_LINE480;
  static_assert(R1 * R2 == 480 && LINE >= R1 * Y2 && LINE >= 480, "the padded stage-1 output fits a line");
};
template <>
struct SrPlan<240> {
  static constexpr int R1 = 15, R2 = 16, Y2 = 17, LINE = 257;
};
template <>
struct SrPlan<256> {
  static constexpr int R1 = 16, R2 = 16, Y2 = 17, LINE = 273;
};
// r06: FftMethod patches of 200 x 200 pixels (the first 5-smooth size past the half-tile kernel's 192) on the same
// tuned transforms -- 200 = 10 x 20: three lines per stage-1 pass (60 of 64 lanes), ten of sixteen lanes per line in stage 2
#ifndef MOF_SR_LINE200  // (A/B: LDS line pitch of the 200-point kernels)
#define MOF_SR_LINE200 211
#endif
template <>
struct SrPlan<200> {
  static constexpr int R1 = 10, R2 = 20, Y2 = 21, LINE = MOF_SR_LINE200;
  static_assert(LINE >= R1 * Y2, "the padded stage-1 output fits a line");
};
// ... and 288 = 16 x 18, 320 = 16 x 20, 384 = 12 x 32 (r06): with the zero padding of the row kernel these serve patches of 271 .. 288, 301 .. 320 and
// 376 .. 384 pixels as well (every radix pair keeps the Nyquist bin of a line free of twiddles: k1 = 0, k2 = R2 / 2)
template <>
struct SrPlan<288> {
  static constexpr int R1 = 16, R2 = 18, Y2 = 19, LINE = 305;
};
template <>
struct SrPlan<320> {
  static constexpr int R1 = 16, R2 = 20, Y2 = 21, LINE = 337;
};
// ... and the sizes whose rows are not whole 8-row waves (RowsReal<N>::TAIL; K7 has its own tail): 270 = 15 x 18, 300 = 15 x 20, 450 = 15 x 30
// (patches of 257 .. 270, 289 .. 300, 433 .. 450 with the padding)
template <>
struct SrPlan<270> {
  static constexpr int R1 = 15, R2 = 18, Y2 = 19, LINE = 285;
};
template <>
struct SrPlan<300> {
  static constexpr int R1 = 15, R2 = 20, Y2 = 21, LINE = 315;
};
template <>
struct SrPlan<450> {
  static constexpr int R1 = 15, R2 = 30, Y2 = 31, LINE = 465;
};
// r06: sizes whose only two-stage plans end in an ODD radix (250 = 10 x 25, 400 = 16 x 25, 432 = 16 x 27): bin N/2 of a line passes twiddles, so
// the four real-only CCS slots would carry rounding noise where the reference's are exact integers -- and C = P / (P^2 + eps) turns a P of 1e-4
// that should be 0 into a bin of magnitude 300. These plans declare NYQ_EXACT = false: the row kernel then also forms the four exact
// integer sums of every image (sum (+-1)^y (+-1)^x p), and the column kernel takes the real-only slots from THEM (SrExactSlots below).
template <>
struct SrPlan<250> {
  static constexpr int R1 = 10, R2 = 25, Y2 = 26, LINE = 261;
  static constexpr bool NYQ_EXACT = false;
};
template <>
struct SrPlan<400> {
  static constexpr int R1 = 16, R2 = 25, Y2 = 26, LINE = 417;
  static constexpr bool NYQ_EXACT = false;
};
template <>
struct SrPlan<432> {
  static constexpr int R1 = 16, R2 = 27, Y2 = 28, LINE = 449;
  static constexpr bool NYQ_EXACT = false;
};
// Row pitch (complex elements) of the transposed row spectra Zh[u][row]: a row workgroup stores 8 rows = 64 bytes per bin u, so the pitch is
// kept a multiple of 8 -- with the plain pitch N the 64-byte pieces of the sizes with N % 8 != 0 (250, 270, 300, 450) straddled two 64-byte
// sectors each and the row kernel ran 3 - 5 x slower on its stores (r06: profiles/r06_zh_pitch_ab.txt)
// (250 -> 256 puts the bins of one store instruction 2 KB apart; 264 instead: -2 %, measured -- the plain round-up stays)
template <int N>
constexpr int sr_zh_pitch() { return (N + 7) & ~7; }
template <class PL, class = void>
struct SrNyqExact { static constexpr bool value = true; };
template <class PL>
struct SrNyqExact<PL, decltype((void)PL::NYQ_EXACT)> { static constexpr bool value = PL::NYQ_EXACT; };
template <>
struct SrPlan<512> {  // 16 x 32 (patches of 501 .. 512 pixels)
  static constexpr int R1 = 16, R2 = 32, Y2 = 33, LINE = 529;
};
template <>
struct SrPlan<360> {  // 15 x 24 (patches of 325 .. 360 pixels): 45 one-wave row workgroups per image, 180 row pairs = 22 candidate workgroups of 8 lines and one of 4
  static constexpr int R1 = 15, R2 = 24, Y2 = 25, LINE = 377;
};
template <>
struct SrPlan<384> {
  static constexpr int R1 = 12, R2 = 32, Y2 = 33, LINE = 397;
};
// r06, below the FFT engine's large-patch band (its own kernels serve these patch sizes; here for the ESTIMATOR at small resolutions)
template <>
struct SrPlan<128> {  // 8 x 16
  static constexpr int R1 = 8, R2 = 16, Y2 = 17, LINE = 137;
};
template <>
struct SrPlan<96> {  // 8 x 12
  static constexpr int R1 = 8, R2 = 12, Y2 = 13, LINE = 105;
};
template <>
struct SrPlan<100> {  // 10 x 10
  static constexpr int R1 = 10, R2 = 10, Y2 = 11, LINE = 111;
};
template <>
struct SrPlan<108> {  // 9 x 12
  static constexpr int R1 = 9, R2 = 12, Y2 = 13, LINE = 117;
};
template <>
struct SrPlan<120> {  // 12 x 10
  static constexpr int R1 = 12, R2 = 10, Y2 = 11, LINE = 133;
};
template <>
struct SrPlan<150> {  // 15 x 10
  static constexpr int R1 = 15, R2 = 10, Y2 = 11, LINE = 165;
};
template <>
struct SrPlan<162> {  // 9 x 18
  static constexpr int R1 = 9, R2 = 18, Y2 = 19, LINE = 171;
};
template <>
struct SrPlan<144> {  // 9 x 16
  static constexpr int R1 = 9, R2 = 16, Y2 = 17, LINE = 153;
};
template <>
struct SrPlan<160> {  // 10 x 16
  static constexpr int R1 = 10, R2 = 16, Y2 = 17, LINE = 171;
};
template <>
struct SrPlan<180> {  // 10 x 18
  static constexpr int R1 = 10, R2 = 18, Y2 = 19, LINE = 191;
};
template <>
struct SrPlan<192> {  // 12 x 16
  static constexpr int R1 = 12, R2 = 16, Y2 = 17, LINE = 205;
};
// r06: the ODD transform sizes getOptimalDFTSize can return in the large-patch band (patches of 217 .. 225, 241 .. 243, 361 .. 375, 401 .. 405,
// 601 .. 625, 649 .. 675, 721 .. 729). Two real rows still share a complex line -- the last row of an image shares its line with a row of zeros --,
// there are (N + 1) / 2 = N / 2 + 1 half-spectrum columns as for even N, no Nyquist bin (only bin (0, 0) is real-only), every bin u > 0 has a
// distinct Hermitian partner N - u. LINE >= N + 1: the last 16-byte piece of a line carries one element past it.
template <>
struct SrPlan<225> {  // 15 x 15
  static constexpr int R1 = 15, R2 = 15, Y2 = 16, LINE = 241;
};
template <>
struct SrPlan<243> {  // 9 x 27
  static constexpr int R1 = 9, R2 = 27, Y2 = 28, LINE = 253;
};
template <>
struct SrPlan<375> {  // 15 x 25
  static constexpr int R1 = 15, R2 = 25, Y2 = 26, LINE = 391;
};
template <>
struct SrPlan<405> {  // 15 x 27
  static constexpr int R1 = 15, R2 = 27, Y2 = 28, LINE = 421;
};
template <>
struct SrPlan<625> {  // 25 x 25
  static constexpr int R1 = 25, R2 = 25, Y2 = 26, LINE = 651;
};
template <>
struct SrPlan<675> {  // 25 x 27
  static constexpr int R1 = 25, R2 = 27, Y2 = 28, LINE = 701;
};
template <>
struct SrPlan<729> {  // 27 x 27
  static constexpr int R1 = 27, R2 = 27, Y2 = 28, LINE = 757;
};
// r06: first radix up to 32 (stage 2 of wave_fft then runs 32 lanes per line, two lines per pass): every even size cv::getOptimalDFTSize
// can return between 512 and 960, and 324 / 486 / 500 below -- all with an even last radix whose Nyquist bin passes no twiddle
// (16, 18 and 30 by decimation in time, 20 / 24 by their Cooley-Tukey split, 32), so the real-only slots stay exact.
template <>
struct SrPlan<324> {  // 18 x 18
  static constexpr int R1 = 18, R2 = 18, Y2 = 19, LINE = 343;
};
template <>
struct SrPlan<486> {  // 27 x 18
  static constexpr int R1 = 27, R2 = 18, Y2 = 19, LINE = 513;
};
template <>
struct SrPlan<500> {  // 25 x 20
  static constexpr int R1 = 25, R2 = 20, Y2 = 21, LINE = 525;
};
template <>
struct SrPlan<540> {  // 27 x 20
  static constexpr int R1 = 27, R2 = 20, Y2 = 21, LINE = 567;
};
template <>
struct SrPlan<576> {  // 18 x 32
  static constexpr int R1 = 18, R2 = 32, Y2 = 33, LINE = 595;
};
template <>
struct SrPlan<600> {  // 25 x 24
  static constexpr int R1 = 25, R2 = 24, Y2 = 25, LINE = 625;
};
template <>
struct SrPlan<640> {  // 20 x 32
  static constexpr int R1 = 20, R2 = 32, Y2 = 33, LINE = 661;
};
template <>
struct SrPlan<648> {  // 27 x 24
  static constexpr int R1 = 27, R2 = 24, Y2 = 25, LINE = 675;
};
template <>
struct SrPlan<720> {  // 24 x 30
  static constexpr int R1 = 24, R2 = 30, Y2 = 31, LINE = 745;
};
template <>
struct SrPlan<750> {  // 25 x 30
  static constexpr int R1 = 25, R2 = 30, Y2 = 31, LINE = 775;
};
template <>
struct SrPlan<768> {  // 24 x 32
  static constexpr int R1 = 24, R2 = 32, Y2 = 33, LINE = 793;
};
template <>
struct SrPlan<800> {  // 25 x 32
  static constexpr int R1 = 25, R2 = 32, Y2 = 33, LINE = 825;
};
template <>
struct SrPlan<810> {  // 27 x 30
  static constexpr int R1 = 27, R2 = 30, Y2 = 31, LINE = 837;
};
template <>
struct SrPlan<864> {  // 27 x 32
  static constexpr int R1 = 27, R2 = 32, Y2 = 33, LINE = 891;
};
template <>
struct SrPlan<900> {  // 30 x 30
  static constexpr int R1 = 30, R2 = 30, Y2 = 31, LINE = 931;
};
template <>
struct SrPlan<960> {  // 30 x 32
  static constexpr int R1 = 30, R2 = 32, Y2 = 33, LINE = 991;
};
template <>
struct SrPlan<216> {  // 12 x 18: three lines per stage-1 pass (54 of 64 lanes), twelve of sixteen lanes per line in stage 2
  static constexpr int R1 = 12, R2 = 18, Y2 = 19, LINE = 229;
};

// Zt / Zh / Dt (and the log-polar images) are STREAMS: written once by one kernel, read once by the next, hundreds of MB per
// pass. Marking those accesses non-temporal keeps them from displacing each other's lines on their way through the caches:
// same-box A/B (r03) c5seq 492 k -> 516 k pairs/s with K5s / K6s alone. MOF_SR_NT=0 builds the plain form.
#ifndef MOF_SR_NT
#define MOF_SR_NT 1
#endif
// MOF_SR_L2_ABLATE=1 (diagnostic build, results wrong by design): every frame's Zh and every pair's Dt live in slot 0 --
// K5s writes, K6s reads and writes, K7 reads the SAME 0.9 MB over and over, so the intermediates stay in the L2s instead of going through
// HBM. What that build gains over the product is the most ANY decomposition that keeps Zh / Dt on chip can return (it pays no hand-off).
#ifndef MOF_SR_L2_ABLATE
#define MOF_SR_L2_ABLATE 0
#endif
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load(const float4* p) {
#if MOF_SR_NT
  const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void stream_store(float4* p, float4 v) {
#if MOF_SR_NT
  const v4f_t w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<v4f_t*>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ uint32_t stream_load(const uint32_t* p) {
#if MOF_SR_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

__device__ __forceinline__ void butterfly20(cf* v) {  // 5 x 4 (pc_plan.hpp: butterfly_ct), twiddles W_20^j = (cos(pi j / 10), -sin(pi j / 10))
  const cf w[13] = {{1.00000000000000000000f, -0.00000000000000000000f}, {0.95105651629515353118f, -0.30901699437494739575f}, {0.80901699437494745126f, -0.58778525229247313710f}, {0.58778525229247313710f, -0.80901699437494745126f}, {0.30901699437494745126f, -0.95105651629515353118f}, {0.00000000000000006123f, -1.00000000000000000000f}, {-0.30901699437494734024f, -0.95105651629515364220f}, {-0.58778525229247302608f, -0.80901699437494745126f}, {-0.80901699437494734024f, -0.58778525229247324813f}, {-0.95105651629515353118f, -0.30901699437494750677f}, {-1.00000000000000000000f, -0.00000000000000012246f}, {-0.95105651629515375323f, 0.30901699437494689615f}, {-0.80901699437494756229f, 0.58778525229247302608f}};
  butterfly_ct<5, 4>(v, w);
}
__device__ __forceinline__ void butterfly18(cf* v) {  // 9 x 2, decimation in time: X[k1 + 9 k2] = DFT9(even)[k1] +- W_18^k1 DFT9(odd)[k1]. Bin 9 (the
  // Nyquist bin of a line) = sum(even) - sum(odd) passes no twiddle: exact on integers, as the real-only CCS slots need (pc_plan_build.hpp)
  const cf w[9] = {{1.00000000000000000000f, -0.00000000000000000000f}, {0.93969262078590842791f, -0.34202014332566871291f}, {0.76604444311897801345f, -0.64278760968653925190f}, {0.50000000000000011102f, -0.86602540378443859659f}, {0.17364817766693041445f, -0.98480775301220802032f}, {-0.17364817766693030343f, -0.98480775301220802032f}, {-0.49999999999999977796f, -0.86602540378443870761f}, {-0.76604444311897790243f, -0.64278760968653947394f}, {-0.93969262078590831688f, -0.34202014332566887944f}};
  cf a[9], b[9];
#pragma unroll
  for (int n1 = 0; n1 < 9; ++n1) {
    a[n1] = v[2 * n1];
    b[n1] = v[2 * n1 + 1];
  }
  butterfly9(a);
  butterfly9(b);
#pragma unroll
  for (int k1 = 0; k1 < 9; ++k1) {
    const cf t = k1 == 0 ? b[0] : cmul(b[k1], w[k1]);
    v[k1] = {a[k1].x + t.x, a[k1].y + t.y};
    v[k1 + 9] = {a[k1].x - t.x, a[k1].y - t.y};
  }
}
// R = RA x RB with a radix-RB decimation in time: RB interleaved sub-sequences through the RA-point butterfly, twiddles W_R^{j k1}, then RB-point
// butterflies across them: X[k1 + RA k2]. (Odd R: no exact Nyquist bin to protect -- SrPlan::NYQ_EXACT = false takes care of the slots.)
__device__ __forceinline__ void butterfly25(cf* v) {  // 5 x 5
  const cf w[17] = {{1.00000000000000000000f, -0.00000000000000000000f}, {0.96858316112863107605f, -0.24868988716485479484f}, {0.87630668004386358394f, -0.48175367410171532345f}, {0.72896862742141155245f, -0.68454710592868861507f}, {0.53582679497899654564f, -0.84432792550201507531f}, {0.30901699437494745126f, -0.95105651629515353118f}, {0.06279051952931352654f, -0.99802672842827155897f}, {-0.18738131458572460097f, -0.98228725072868872115f}, {-0.42577929156507271502f, -0.90482705246601946580f}, {-0.63742398974868974548f, -0.77051324277578925326f}, {-0.80901699437494734024f, -0.58778525229247324813f}, {-0.92977648588825134723f, -0.36812455268467814129f}, {-0.99211470131447776488f, -0.12533323356430453588f}, {-0.99211470131447787590f, 0.12533323356430428608f}, {-0.92977648588825145826f, 0.36812455268467791925f}, {-0.80901699437494778433f, 0.58778525229247269301f}, {-0.63742398974868952344f, 0.77051324277578936428f}};
  cf a[5][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
#pragma unroll
    for (int n1 = 0; n1 < 5; ++n1) a[j][n1] = v[5 * n1 + j];
    butterfly5(a[j]);
  }
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    cf t[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) t[j] = (j * k1 == 0) ? a[j][k1] : cmul(a[j][k1], w[j * k1]);
    butterfly5(t);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[k1 + 5 * k2] = t[k2];
  }
}
__device__ __forceinline__ void butterfly27(cf* v) {  // 9 x 3
  const cf w[17] = {{1.00000000000000000000f, -0.00000000000000000000f}, {0.97304487057982380627f, -0.23061587074244016549f}, {0.89363264032341227505f, -0.44879918020046216665f}, {0.76604444311897801345f, -0.64278760968653925190f}, {0.59715859170278617896f, -0.80212319275504373461f}, {0.39607976603915689973f, -0.91821610688027399672f}, {0.17364817766693041445f, -0.98480775301220802032f}, {-0.05814482891047577373f, -0.99830815827126817563f}, {-0.28680323271109020578f, -0.95798951231548890028f}, {-0.49999999999999977796f, -0.86602540378443870761f}, {-0.68624163786873348947f, -0.72737364157304884582f}, {-0.83548781141293626540f, -0.54950897807080623103f}, {-0.93969262078590831688f, -0.34202014332566887944f}, {-0.99323835774194302317f, -0.11609291412522992903f}, {-0.99323835774194302317f, 0.11609291412523012332f}, {-0.93969262078590853893f, 0.34202014332566821331f}, {-0.83548781141293648744f, 0.54950897807080600899f}};
  cf a[3][9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int n1 = 0; n1 < 9; ++n1) a[j][n1] = v[3 * n1 + j];
    butterfly9(a[j]);
  }
#pragma unroll
  for (int k1 = 0; k1 < 9; ++k1) {
    cf t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] = (j * k1 == 0) ? a[j][k1] : cmul(a[j][k1], w[j * k1]);
    butterfly3(t);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) v[k1 + 9 * k2] = t[k2];
  }
}
__device__ __forceinline__ void butterfly30(cf* v) {  // 15 x 2, decimation in time (as butterfly18): bin 15 = sum(even) - sum(odd), no twiddle
  const cf w[15] = {{1.00000000000000000000f, -0.00000000000000000000f}, {0.97814760073380568883f, -0.20791169081775931482f}, {0.91354545764260086660f, -0.40673664307580015276f}, {0.80901699437494745126f, -0.58778525229247313710f}, {0.66913060635885823757f, -0.74314482547739413310f}, {0.50000000000000011102f, -0.86602540378443859659f}, {0.30901699437494745126f, -0.95105651629515353118f}, {0.10452846326765345697f, -0.99452189536827328986f}, {-0.10452846326765333207f, -0.99452189536827340088f}, {-0.30901699437494734024f, -0.95105651629515364220f}, {-0.49999999999999977796f, -0.86602540378443870761f}, {-0.66913060635885790450f, -0.74314482547739446616f}, {-0.80901699437494734024f, -0.58778525229247324813f}, {-0.91354545764260097762f, -0.40673664307580004174f}, {-0.97814760073380568883f, -0.20791169081775931482f}};
  cf a[15], b[15];
#pragma unroll
  for (int n1 = 0; n1 < 15; ++n1) {
    a[n1] = v[2 * n1];
    b[n1] = v[2 * n1 + 1];
  }
  butterfly15(a);
  butterfly15(b);
#pragma unroll
  for (int k1 = 0; k1 < 15; ++k1) {
    const cf t = k1 == 0 ? b[0] : cmul(b[k1], w[k1]);
    v[k1] = {a[k1].x + t.x, a[k1].y + t.y};
    v[k1 + 15] = {a[k1].x - t.x, a[k1].y - t.y};
  }
}
__device__ __forceinline__ void butterfly24(cf* v) {  // 3 x 8 (pc_plan.hpp: butterfly_ct), twiddles W_24^j; bin 12 = (k1 = 0, k2 = 4) passes no twiddle
  const cf w[15] = {{1.00000000000000000000f, -0.00000000000000000000f}, {0.96592582628906831221f, -0.25881904510252073948f}, {0.86602540378443870761f, -0.49999999999999994449f}, {0.70710678118654757274f, -0.70710678118654746172f}, {0.50000000000000011102f, -0.86602540378443859659f}, {0.25881904510252073948f, -0.96592582628906831221f}, {0.00000000000000006123f, -1.00000000000000000000f}, {-0.25881904510252062845f, -0.96592582628906831221f}, {-0.49999999999999977796f, -0.86602540378443870761f}, {-0.70710678118654746172f, -0.70710678118654757274f}, {-0.86602540378443870761f, -0.49999999999999994449f}, {-0.96592582628906820119f, -0.25881904510252101703f}, {-1.00000000000000000000f, -0.00000000000000012246f}, {-0.96592582628906831221f, 0.25881904510252079499f}, {-0.86602540378443881863f, 0.49999999999999972244f}};
  butterfly_ct<3, 8>(v, w);
}
template <int R>
__device__ __forceinline__ void bfly(cf* v) {
  if constexpr (R == 15) butterfly15(v);
  else if constexpr (R == 32) butterfly32(v);
  else if constexpr (R == 30) butterfly30(v);
  else if constexpr (R == 27) butterfly27(v);
  else if constexpr (R == 25) butterfly25(v);
  else if constexpr (R == 24) butterfly24(v);
  else if constexpr (R == 20) butterfly20(v);
  else if constexpr (R == 18) butterfly18(v);
  else if constexpr (R == 12) butterfly12(v);
  else if constexpr (R == 10) butterfly10(v);
  else if constexpr (R == 9) butterfly9(v);
  else butterfly<R>(v);
}

constexpr int SR_LINES = 8;      // lines per workgroup in K5 / K7 (8 rows, 8 row pairs) and in K6 (4 columns + 4 mirrors)
constexpr int SR_T = 128;        // two waves, four lines each
constexpr int COLS_CW = SR_LINES / 2;

// Inter-stage twiddles of the lane's stage-1 slot: W_N^{n2 k1}, k1 = 1..R1-1, n2 = lane % R2 (the same for every line
// the lane ever transforms, so they are fetched once per kernel).
template <int N>
struct SrTw {
  cf w[SrPlan<N>::R1 - 1];
  __device__ __forceinline__ void load(const float* __restrict__ table, int lane) {
    const int n2 = lane % SrPlan<N>::R2;
#pragma unroll
    for (int k1 = 1; k1 < SrPlan<N>::R1; ++k1) {
      const float2 t = *reinterpret_cast<const float2*>(table + 2 * (n2 * k1));  // n2 * k1 < N
      w[k1 - 1] = {t.x, t.y};
    }
  }
};

// Forward DFT of `nl` (2 or 4) lines of length N owned by ONE wave, in place, natural order in and out.
//   stage 1: lane = (line, n2): radix R1 over x[R2 n1 + n2], twiddle W_N^{n2 k1}, stored at y[Y2 k1 + n2]
//   stage 2: lane = (line, k1): radix R2 over y[Y2 k1 + n2], X[k1 + R1 k2] stored in natural order (or handed to
//            `sink(line, k1, v)` instead when the caller consumes the result from registers).
// A wave's LDS instructions execute in order and every lane reads all its inputs before it writes, so in place is safe.
template <int N, class Sink>
__device__ __forceinline__ void wave_fft(cf* __restrict__ z, int nl, int lane, const SrTw<N>& tw, Sink sink) {
  using P = SrPlan<N>;
  constexpr int LP1 = 64 / P::R2;  // lines per stage-1 pass
  for (int l0 = 0; l0 < nl; l0 += LP1) {
    const int l = l0 + lane / P::R2, n2 = lane % P::R2;
    if (l < nl && lane < LP1 * P::R2) {  // (R2 = 16 packs four lines into a pass; a two-line call leaves half the wave idle; R2 = 20: lanes 60 .. 63 idle)
      cf* line = z + l * P::LINE;
      cf v[P::R1];
#pragma unroll
      for (int n1 = 0; n1 < P::R1; ++n1) v[n1] = lds_read(&line[P::R2 * n1 + n2]);
      bfly<P::R1>(v);
      line[n2] = v[0];
#pragma unroll
      for (int k1 = 1; k1 < P::R1; ++k1) line[P::Y2 * k1 + n2] = cmul(v[k1], tw.w[k1 - 1]);
    }
  }
  wave_sync();
  if constexpr (P::R1 <= 16) {
    // 16 lanes per line (15 of them active when R1 = 15); nl = 2 leaves the upper half of the wave idle
    const int l = lane >> 4, k1 = lane & 15;
    const bool on = l < nl && k1 < P::R1;
    cf* line = z + l * P::LINE;
    cf v[P::R2];
    if (on) {
#pragma unroll
      for (int n2 = 0; n2 < P::R2; ++n2) v[n2] = lds_read(&line[P::Y2 * k1 + n2]);
      bfly<P::R2>(v);
      sink(line, l, k1, v);
    }
  } else {
    // R1 = 17 .. 32 (r06): 32 lanes per line, two lines per pass (a pass reads and rewrites its own two lines only)
    static_assert(P::R1 <= 32, "one lane per first-stage bin");
    for (int l0 = 0; l0 < nl; l0 += 2) {
      const int l = l0 + (lane >> 5), k1 = lane & 31;
      const bool on = l < nl && k1 < P::R1;
      cf* line = z + l * P::LINE;
      cf v[P::R2];
      if (on) {
#pragma unroll
        for (int n2 = 0; n2 < P::R2; ++n2) v[n2] = lds_read(&line[P::Y2 * k1 + n2]);
        bfly<P::R2>(v);
        sink(line, l, k1, v);
      }
    }
  }
  wave_sync();
}

// default sink: natural-order store
template <int N>
struct StoreNatural {
  __device__ __forceinline__ void operator()(cf* line, int, int k1, const cf* v) const {
#pragma unroll
    for (int k2 = 0; k2 < SrPlan<N>::R2; ++k2) line[k1 + SrPlan<N>::R1 * k2] = v[k2];
  }
};

// compile-time loop: the body sees its index as an integral_constant
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

}  // namespace

}  // namespace mof
