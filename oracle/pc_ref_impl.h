/*
 * pc_ref_impl.h -- body of the phase-correlation oracle, instantiated twice
 * (R = float, R = double) by pc_ref.c. TEST INFRASTRUCTURE ONLY; parity unpinned
 * (see oracle.h). Every "ref:" tag cites /root/reference/src/FftMethod.cpp.
 *
 * The spectrum is held as a full N x N complex array, but only the bins that
 * OpenCV's CCS-packed real spectrum stores are ever computed on; the rest is
 * filled by Hermitian symmetry before the inverse transform. Which arithmetic
 * flavour a bin gets follows where CCS stores it (N even; for odd N there is no
 * N/2 column / row and the only real-only slot is DC):
 *   - column-frequency 0 and N/2, row-frequency 0 and N/2 : the 4 real-only slots
 *   - column-frequency 0 and N/2, row-frequency 1..N/2-1  : "column" pairs (double maths)
 *   - column-frequency 1..N/2-1, every row-frequency      : "interior" pairs (working-type maths)
 * (ref: magSpectrums :103-131, divSpectrums :1123-1181).
 * N is the PADDED size getOptimalDFTSize(n) -- cv::phaseCorrelate zero-pads to it.
 */

#ifndef R
#error "define R, SUFFIX before including"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef struct { R re, im; } FN(cpx);

/* Twiddles are computed in double and stored in the working type, as OpenCV's
 * DFT does for CV_32F. tw[j] = exp(-2*pi*i*j/n); the axis values are patched to
 * be exact so constant inputs give exactly-zero AC bins on power-of-two sizes. */
static void FN(make_twiddles)(FN(cpx) * tw, int n) {
  for (int j = 0; j < n; ++j) {
    double ang = -2.0 * 3.14159265358979323846 * (double)j / (double)n;
    double c = cos(ang), s = sin(ang);
    if ((4 * j) % n == 0) {
      int q = (4 * j) / n; /* quarter turns */
      c = (q == 0) ? 1.0 : (q == 2) ? -1.0 : 0.0;
      s = (q == 1) ? -1.0 : (q == 3) ? 1.0 : 0.0;
    }
    tw[j].re = (R)c;
    tw[j].im = (R)s;
  }
}

static int FN(smallest_factor)(int n) {
  if (n % 2 == 0) return 2;
  for (int p = 3; p * p <= n; p += 2)
    if (n % p == 0) return p;
  return n;
}

/* Recursive decimation-in-time mixed-radix transform; inverse uses conjugated
 * twiddles and is unscaled (OpenCV idft without DFT_SCALE, ref :1497). */
static void FN(fft_rec)(const FN(cpx) * in, int stride, FN(cpx) * out, int n, const FN(cpx) * tw, int tw_n, int inverse) {
  if (n == 1) {
    out[0] = in[0];
    return;
  }
  int p = FN(smallest_factor)(n);
  int m = n / p;
  for (int r = 0; r < p; ++r) FN(fft_rec)(in + (size_t)r * stride, stride * p, out + (size_t)r * m, m, tw, tw_n, inverse);
  int tw_step = tw_n / n;
  FN(cpx) t[64];
  for (int k = 0; k < m; ++k) {
    for (int r = 0; r < p; ++r) {
      FN(cpx) w = tw[((size_t)r * k * tw_step) % tw_n];
      if (inverse) w.im = -w.im;
      FN(cpx) v = out[(size_t)r * m + k];
      if (r == 0 || k == 0) {
        t[r] = v;
      } else {
        t[r].re = v.re * w.re - v.im * w.im;
        t[r].im = v.re * w.im + v.im * w.re;
      }
    }
    if (p == 2) {
      out[k].re = t[0].re + t[1].re;
      out[k].im = t[0].im + t[1].im;
      out[m + k].re = t[0].re - t[1].re;
      out[m + k].im = t[0].im - t[1].im;
    } else {
      for (int q = 0; q < p; ++q) {
        R sr = t[0].re, si = t[0].im;
        for (int r = 1; r < p; ++r) {
          FN(cpx) w = tw[(size_t)((r * q) % p) * (tw_n / p)];
          if (inverse) w.im = -w.im;
          sr += t[r].re * w.re - t[r].im * w.im;
          si += t[r].re * w.im + t[r].im * w.re;
        }
        out[(size_t)q * m + k].re = sr;
        out[(size_t)q * m + k].im = si;
      }
    }
  }
}

/* In-place 2-D transform of an n x n complex array: rows, then columns. */
static void FN(fft2d)(FN(cpx) * data, int n, const FN(cpx) * tw, int inverse, FN(cpx) * scratch) {
  for (int y = 0; y < n; ++y) {
    FN(fft_rec)(data + (size_t)y * n, 1, scratch, n, tw, n, inverse);
    memcpy(data + (size_t)y * n, scratch, sizeof(FN(cpx)) * (size_t)n);
  }
  for (int x = 0; x < n; ++x) {
    FN(fft_rec)(data + x, n, scratch, n, tw, n, inverse);
    for (int y = 0; y < n; ++y) data[(size_t)y * n + x] = scratch[y];
  }
}

/* One CCS bin of the normalised cross-power spectrum C = divSpectrums(P, magSpectrums(P)),
 * P = mulSpectrums(A, B, conjB=true). kind: 0 real-only slot, 1 column pair, 2 interior pair. */
static FN(cpx) FN(cross_power_bin)(FN(cpx) a, FN(cpx) b, int kind) {
  const R eps = (R)FLT_EPSILON; /* ref :1117 (the f64 instantiation keeps the f32 path's eps) */
  FN(cpx) c;
  if (kind == 0) {
    /* real-only slots: plain product (cv::mulSpectrums), SQUARE as "magnitude"
     * (ref :107-109), A/(B+eps) (ref :1127-1129) -- SURVEY F8 quirk. */
    R p = a.re * b.re;
    R pm = p * p;
    c.re = p / (pm + eps);
    c.im = (R)0;
    return c;
  }
  /* cv::mulSpectrums with conjB: products formed in double, stored in the working type. */
  R pre = (R)((double)a.re * (double)b.re + (double)a.im * (double)b.im);
  R pim = (R)((double)a.im * (double)b.re - (double)a.re * (double)b.im);
  /* magSpectrums: sqrt in double, stored at the real slot; imag slot 0 (ref :85, :112-113, :129). */
  R mag = (R)sqrt((double)pre * (double)pre + (double)pim * (double)pim);
  const R zero = (R)0;
  if (kind == 1) {
    /* ref :1131-1140 -- first/last CCS column, evaluated in double. */
    double denom = (double)mag * (double)mag + (double)zero * (double)zero + (double)eps;
    double re = (double)pre * (double)mag + (double)pim * (double)zero;
    double im = (double)pim * (double)mag - (double)pre * (double)zero;
    c.re = (R)(re / denom);
    c.im = (R)(im / denom);
  } else {
    /* ref :1166-1172 -- interior pairs: working-type expression, cast, divide in double. */
    double denom = (double)(R)((R)((R)(mag * mag) + (R)(zero * zero)) + eps);
    double re = (double)(R)((R)(pre * mag) + (R)(pim * zero));
    double im = (double)(R)((R)(pim * mag) - (R)(pre * zero));
    c.re = (R)(re / denom);
    c.im = (R)(im / denom);
  }
  return c;
}

int FN(oracle_phase_correlate)(const R* a, size_t a_stride, const R* b, size_t b_stride, int n_in, double* out_xy,
                               oracle_pc_diag* diag, R* surface) {
  if (!a || !b || !out_xy || n_in < 2) return -1;
  /* cv::phaseCorrelate pads both images with zeros (copyMakeBorder, BORDER_CONSTANT 0, bottom / right) to
   * M = N = getOptimalDFTSize(rows / cols) before anything else [published OpenCV algorithm, unpinned]; everything from
   * here on -- spectra, fftShift, peak, centroid, `center` -- lives on the PADDED n x n image. n may be odd (e.g. 74 -> 75). */
  const int n = oracle_optimal_dft_size(n_in);
  if (n < 2) return -1;
  const size_t nn = (size_t)n * n;
  FN(cpx)* A = (FN(cpx)*)malloc(sizeof(FN(cpx)) * nn);
  FN(cpx)* B = (FN(cpx)*)malloc(sizeof(FN(cpx)) * nn);
  FN(cpx)* C = (FN(cpx)*)malloc(sizeof(FN(cpx)) * nn);
  FN(cpx)* tw = (FN(cpx)*)malloc(sizeof(FN(cpx)) * (size_t)n);
  FN(cpx)* scratch = (FN(cpx)*)malloc(sizeof(FN(cpx)) * (size_t)n);
  R* S = (R*)malloc(sizeof(R) * nn);
  if (!A || !B || !C || !tw || !scratch || !S) {
    free(A); free(B); free(C); free(tw); free(scratch); free(S);
    return -2;
  }
  FN(make_twiddles)(tw, n);

  /* dft(a, DFT_REAL_OUTPUT), dft(b, ...) -- ref :1491-1493. No window. */
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      const int inside = y < n_in && x < n_in;
      A[(size_t)y * n + x].re = inside ? a[(size_t)y * a_stride + x] : (R)0;
      A[(size_t)y * n + x].im = (R)0;
      B[(size_t)y * n + x].re = inside ? b[(size_t)y * b_stride + x] : (R)0;
      B[(size_t)y * n + x].im = (R)0;
    }
  FN(fft2d)(A, n, tw, 0, scratch);
  FN(fft2d)(B, n, tw, 0, scratch);

  /* mulSpectrums(conjB) -> magSpectrums -> divSpectrums on the CCS bins (ref :1494-1496). CCS of an n x n real transform:
   * column-frequency 0 (and n/2 when n is even) is stored down the first (last) column as Re Y[0], (Re, Im) Y[1..], and --
   * when n is even -- Re Y[n/2]: `for (k = 0; k < (cols % 2 ? 1 : 2); k++)`, `if (rows % 2 == 0)`, `for (j = 1; j <= rows - 2;
   * j += 2)` (ref :103-119, :1123-1140); the other column-frequencies 1 .. (n-1)/2 are (Re, Im) pairs in every row
   * (`for (j = j0; j < j1; j += 2)`, j1 = ncols - (cols % 2 == 0), ref :96-97, :121-131). */
  const int h = n / 2; /* floor */
  const int even = (n % 2 == 0);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c <= h; ++c) {
      const int edge_col = (c == 0 || (even && c == h));
      if (edge_col && r > h) continue; /* not stored in CCS; filled by symmetry below */
      int kind = edge_col ? ((r == 0 || (even && r == h)) ? 0 : 1) : 2;
      FN(cpx) av = A[(size_t)r * n + c], bv = B[(size_t)r * n + c];
      if (kind == 0) { av.im = (R)0; bv.im = (R)0; } /* CCS holds only the real part there */
      C[(size_t)r * n + c] = FN(cross_power_bin)(av, bv, kind);
    }
  /* Hermitian fill: C[N-r][N-c] = conj(C[r][c]). */
  for (int c = 0; c <= h; c += (h > 0 ? h : 1)) {
    if (c == h && !even) break; /* odd n: only column-frequency 0 is an edge column */
    for (int r = h + 1; r < n; ++r) {
      C[(size_t)r * n + c].re = C[(size_t)(n - r) * n + c].re;
      C[(size_t)r * n + c].im = -C[(size_t)(n - r) * n + c].im;
    }
  }
  for (int r = 0; r < n; ++r)
    for (int c = h + 1; c < n; ++c) {
      FN(cpx) v = C[(size_t)((n - r) % n) * n + (n - c)];
      C[(size_t)r * n + c].re = v.re;
      C[(size_t)r * n + c].im = -v.im;
    }

  /* idft, unscaled, real output (ref :1497); fftShift (ref :1257-1323): even sizes swap q0<->q3, q1<->q2 (:1297-1305), odd
   * sizes move the (xMid + 1)-wide first block behind the xMid-wide second one (:1306-1317) -- either way source index i
   * lands at (i + n/2) mod n with n/2 = xMid = n >> 1. */
  FN(fft2d)(C, n, tw, 1, scratch);
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) S[(size_t)((y + h) % n) * n + ((x + h) % n)] = C[(size_t)y * n + x].re;

  /* minMaxLoc: first maximum in row-major order (ref :1539). */
  int px = 0, py = 0;
  R best = S[0];
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x)
      if (S[(size_t)y * n + x] > best) {
        best = S[(size_t)y * n + x];
        px = x;
        py = y;
      }

  /* weightedCentroid(C, peak, Size(5,5)) -- ref :1337-1383. */
  int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
  if (minr < 0) minr = 0;
  if (minc < 0) minc = 0;
  if (maxr > n - 1) maxr = n - 1;
  if (maxc > n - 1) maxc = n - 1;
  double cx = 0.0, cy = 0.0, sum = 0.0;
  for (int y = minr; y <= maxr; ++y)
    for (int x = minc; x <= maxc; ++x) {
      double v = (double)S[(size_t)y * n + x];
      cx += (double)x * v;
      cy += (double)y * v;
      sum += v;
    }
  double response = sum;
  sum += DBL_EPSILON; /* ref :1378 */
  cx /= sum;
  cy /= sum;

  /* center - t with center = (padded cols / 2.0, padded rows / 2.0) (cv::phaseCorrelate's return value; the caller
   * negates, ref :1836). For odd n the centre is a half-integer while the unshifted origin lands on the integer n >> 1:
   * identical images then give (0.5, 0.5) -- OpenCV's behaviour, reproduced. */
  out_xy[0] = (double)n / 2.0 - cx;
  out_xy[1] = (double)n / 2.0 - cy;

  if (diag) {
    double second = -HUGE_VAL;
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x) {
        if (y >= minr && y <= maxr && x >= minc && x <= maxc) continue;
        if ((double)S[(size_t)y * n + x] > second) second = (double)S[(size_t)y * n + x];
      }
    diag->peak_x = px;
    diag->peak_y = py;
    diag->peak_value = (double)best;
    diag->second_value = second;
    diag->response = response / ((double)n * (double)n);
  }
  if (surface) memcpy(surface, S, sizeof(R) * nn);
  free(A); free(B); free(C); free(tw); free(scratch); free(S);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * The useOCL=true peak model, restated from /root/reference/cl/FftMethod.cl as
 * the maths its kernel chain intends (the kernel synchronises work-groups with barrier(), which
 * OpenCL does not guarantee, so its literal output is not defined; this is the race-free reading):
 *   cross-power   v = A*conj(B) * rsqrt(|A*conj(B)|^2 + FLT_EPSILON), products as mad()   cl:971-982, :1041, :1080
 *                 the four real-only CCS slots get 1/(a*b)                                 cl:1024-1031
 *   inverse       scaled by 1/(N*N)                                                        cl:733, :826
 *   search mask   rows/columns with SEARCH_RADIUS < index < N - SEARCH_RADIUS (unshifted)
 *                 are written as 0                                                         cl:737-746, :823-826
 *   fft shift     index +- N/2 on write                                                    cl:738, :819-826
 *   arg-max       first maximum in row-major order, from -FLT_MAX                          cl:1164-1313
 *   refine        radius 3 (cl:1478): window clipped to the patch, only values > 0, float sums
 *                 over ABSOLUTE frame coordinates, sum seeded with FLT_EPSILON; result
 *                 centroid - (origin + N/2)                                                cl:1315-1379
 * The host takes that value as the shift without negating it (FftMethod.cpp:1544, :1833).
 * out_xy receives the shift (NOT center - t as the OpenCV-model function above). */
int FN(oracle_phase_correlate_ocl)(const R* a, size_t a_stride, const R* b, size_t b_stride, int n, int x0, int y0,
                                   int search_radius, double* out_xy, oracle_pc_diag* diag, R* surface) {
  if (!a || !b || !out_xy || n < 8 || (n & 1) || largest_prime_factor(n) > 61 || search_radius < 0) return -1;
  const size_t nn = (size_t)n * n;
  FN(cpx)* A = (FN(cpx)*)malloc(sizeof(FN(cpx)) * nn);
  FN(cpx)* B = (FN(cpx)*)malloc(sizeof(FN(cpx)) * nn);
  FN(cpx)* C = (FN(cpx)*)malloc(sizeof(FN(cpx)) * nn);
  FN(cpx)* tw = (FN(cpx)*)malloc(sizeof(FN(cpx)) * (size_t)n);
  FN(cpx)* scratch = (FN(cpx)*)malloc(sizeof(FN(cpx)) * (size_t)n);
  R* S = (R*)malloc(sizeof(R) * nn);
  if (!A || !B || !C || !tw || !scratch || !S) {
    free(A); free(B); free(C); free(tw); free(scratch); free(S);
    return -2;
  }
  FN(make_twiddles)(tw, n);
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      A[(size_t)y * n + x].re = a[(size_t)y * a_stride + x];
      A[(size_t)y * n + x].im = (R)0;
      B[(size_t)y * n + x].re = b[(size_t)y * b_stride + x];
      B[(size_t)y * n + x].im = (R)0;
    }
  FN(fft2d)(A, n, tw, 0, scratch);
  FN(fft2d)(B, n, tw, 0, scratch);

  const int h = n / 2;
  const R eps = (R)FLT_EPSILON;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c <= h; ++c) {
      const int edge_col = (c == 0 || c == h);
      if (edge_col && r > h) continue;
      FN(cpx) av = A[(size_t)r * n + c], bv = B[(size_t)r * n + c], v;
      if (edge_col && (r == 0 || r == h)) {
        v.re = (R)1 / (av.re * bv.re); /* cl:1029 */
        v.im = (R)0;
      } else {
        /* cmulnormf(a, conjf(b)): conj b, then mad(a.x,b.x,-a.y*b.y), mad(a.x,b.y,a.y*b.x)  cl:976-982 */
        const R bx = bv.re, by = -bv.im;
        const R mx = FMA_R(av.re, bx, -(av.im * by));
        const R my = FMA_R(av.re, by, av.im * bx);
        const R den = (R)1 / SQRT_R(FMA_R(mx, mx, my * my + eps));
        v.re = mx * den;
        v.im = my * den;
      }
      C[(size_t)r * n + c] = v;
    }
  for (int c = 0; c <= h; c += h)
    for (int r = h + 1; r < n; ++r) {
      C[(size_t)r * n + c].re = C[(size_t)(n - r) * n + c].re;
      C[(size_t)r * n + c].im = -C[(size_t)(n - r) * n + c].im;
    }
  for (int r = 0; r < n; ++r)
    for (int c = h + 1; c < n; ++c) {
      FN(cpx) v = C[(size_t)((n - r) % n) * n + (n - c)];
      C[(size_t)r * n + c].re = v.re;
      C[(size_t)r * n + c].im = -v.im;
    }
  FN(fft2d)(C, n, tw, 1, scratch);

  const R scale = (R)1 / (R)(n * n);
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      const int masked = (y > search_radius && y < n - search_radius) || (x > search_radius && x < n - search_radius);
      const int sy = y < h ? y + h : y - h, sx = x < h ? x + h : x - h;
      S[(size_t)sy * n + sx] = masked ? (R)0 : C[(size_t)y * n + x].re * scale;
    }

  int px = 0, py = 0;
  R best = -(R)FLT_MAX;
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x)
      if (best < S[(size_t)y * n + x]) {
        best = S[(size_t)y * n + x];
        px = x;
        py = y;
      }

  const int radius = 3;
  const int xmin = px - radius >= 0 ? px - radius : 0, xmax = px + radius < n ? px + radius : n - 1;
  const int ymin = py - radius >= 0 ? py - radius : 0, ymax = py + radius < n ? py + radius : n - 1;
  R cx = (R)0, cy = (R)0, sum = eps;
  for (int y = ymin; y <= ymax; ++y)
    for (int x = xmin; x <= xmax; ++x) {
      const R v = S[(size_t)y * n + x];
      if (v > (R)0) {
        cx += (R)(x0 + x) * v;
        cy += (R)(y0 + y) * v;
        sum += v;
      }
    }
  cx /= sum;
  cy /= sum;
  out_xy[0] = (double)(R)(cx - (R)(x0 + h));
  out_xy[1] = (double)(R)(cy - (R)(y0 + h));

  if (diag) {
    double second = -HUGE_VAL;
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x) {
        if (y >= ymin && y <= ymax && x >= xmin && x <= xmax) continue;
        if ((double)S[(size_t)y * n + x] > second) second = (double)S[(size_t)y * n + x];
      }
    diag->peak_x = px;
    diag->peak_y = py;
    diag->peak_value = (double)best;
    diag->second_value = second;
    diag->response = (double)sum;
  }
  if (surface) memcpy(surface, S, sizeof(R) * nn);
  free(A); free(B); free(C); free(tw); free(scratch); free(S);
  return 0;
}

#undef FN
#undef CAT
#undef CAT_
