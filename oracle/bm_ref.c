/*
 * bm_ref.c -- CPU restatement of the reference's two block-matching paths
 * (integer stage). TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h):
 * both classes are dead code in the reference and there are no
 * fixtures; this file follows their text.
 *
 *   BlockMethod::processImage     /root/reference/src/BlockMethod.cpp:25-94
 *   OptFlow_C1_D0, Histogram_C1_D0 /root/reference/src/FastSpacedBMMethod.cl:4-84, :86-169
 *   host grid maths               /root/reference/src/FastSpacedBMMethod_OCL.cpp:81-97, :172-175
 *
 * Both share one geometry: the current block sits at (bx*S + r, by*S + r), the
 * previous-frame search window starts at (bx*S, by*S) and candidate (xs,ys) in
 * [0,2r]^2 compares against prev(bx*S + xs + i, by*S + ys + j); BlockMethod is
 * the case S = samplePointSize (BlockMethod.cpp:45, :53-55 with j,i in [-r,r]).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

void oracle_bm_config_block_method(oracle_bm_config* c, int frame_size, int block, int radius) {
  memset(c, 0, sizeof(*c));
  c->width = frame_size;
  c->height = frame_size;
  c->block = block;
  c->step = 0;
  c->radius = radius;
  c->grid_x = c->grid_y = (frame_size - radius * 2) / block; /* BlockMethod.cpp:11 maxSamplesSide */
  c->low_contrast_rule = 0;
}

void oracle_bm_config_fast_spaced(oracle_bm_config* c, int width, int height, int block, int step, int radius) {
  memset(c, 0, sizeof(*c));
  c->width = width;
  c->height = height;
  c->block = block;
  c->step = step;
  c->radius = radius;
  c->grid_x = (width - radius * 2) / (block + step); /* FastSpacedBMMethod_OCL.cpp:90 */
  c->grid_y = (height - radius * 2) / (block + step);
  c->low_contrast_rule = 1;
}

/* First-maximum histogram mode: std::max_element (BlockMethod.cpp:75-76) and the
 * stable descending bubble sort's element 0 (FastSpacedBMMethod.cl:126-151) agree. */
static int histogram_mode(const int* hist, int bins) {
  int best = 0;
  for (int i = 1; i < bins; ++i)
    if (hist[i] > hist[best]) best = i;
  return best;
}

int oracle_bm_process_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_bm_config* cfg, int8_t* dx,
                         int8_t* dy, int8_t* mode_xy, int32_t* sad_min, int32_t* sad_all) {
  if (!cur || !prev || !cfg || !dx || !dy) return -1;
  const int r = cfg->radius, sps = cfg->block, S = cfg->block + cfg->step, D = 2 * r + 1;
  if (r < 0 || r > 63 || sps < 1 || cfg->step < 0 || cfg->grid_x < 1 || cfg->grid_y < 1) return -1;
  /* every read stays inside the frame: last block's window ends at (g-1)*S + sps + 2r */
  if ((cfg->grid_x - 1) * S + sps + 2 * r > cfg->width || (cfg->grid_y - 1) * S + sps + 2 * r > cfg->height) return -1;

  int* xhist = (int*)calloc((size_t)D, sizeof(int));
  int* yhist = (int*)calloc((size_t)D, sizeof(int));
  int32_t* sad = (int32_t*)malloc(sizeof(int32_t) * (size_t)D * D);
  if (!xhist || !yhist || !sad) { free(xhist); free(yhist); free(sad); return -2; }

  for (int by = 0; by < cfg->grid_y; ++by)
    for (int bx = 0; bx < cfg->grid_x; ++bx) {
      const int cx0 = bx * S + r, cy0 = by * S + r; /* current block origin */
      const int wx0 = bx * S, wy0 = by * S;         /* previous-frame window origin */
      for (int ys = 0; ys < D; ++ys)
        for (int xs = 0; xs < D; ++xs) {
          int32_t acc = 0;
          for (int j = 0; j < sps; ++j) {
            const uint8_t* pc = cur + (size_t)(cy0 + j) * pitch + cx0;
            const uint8_t* pp = prev + (size_t)(wy0 + ys + j) * pitch + wx0 + xs;
            for (int i = 0; i < sps; ++i) {
              int d = (int)pc[i] - (int)pp[i];
              acc += d < 0 ? -d : d;
            }
          }
          sad[ys * D + xs] = acc;
        }
      /* arg-min, first occurrence in row-major order: cv::minMaxLoc (BlockMethod.cpp:63);
       * per-row strict '>' then strict '>' across rows (FastSpacedBMMethod.cl:50-56, :66-73). */
      int best = 0;
      for (int k = 1; k < D * D; ++k)
        if (sad[k] < sad[best]) best = k;
      int mx = best % D, my = best / D;
      int32_t minval = sad[best];
      /* low-contrast rule: (abssum[r][r] - min) <= r*r*0.2 -> (0,0)  (FastSpacedBMMethod.cl:2, :77-82) */
      if (cfg->low_contrast_rule && (double)(sad[r * D + r] - minval) <= (double)(r * r) * 0.2) {
        mx = r;
        my = r;
      }
      const int b = by * cfg->grid_x + bx;
      dx[b] = (int8_t)(mx - r);
      dy[b] = (int8_t)(my - r);
      xhist[mx]++; /* BlockMethod.cpp:65-66; FastSpacedBMMethod.cl:118-122 */
      yhist[my]++;
      if (sad_min) sad_min[b] = minval;
      if (sad_all) memcpy(sad_all + (size_t)b * D * D, sad, sizeof(int32_t) * (size_t)D * D);
    }
  if (mode_xy) {
    mode_xy[0] = (int8_t)(histogram_mode(xhist, D) - r);
    mode_xy[1] = (int8_t)(histogram_mode(yhist, D) - r);
  }
  free(xhist); free(yhist); free(sad);
  return 0;
}

int oracle_bm_histogram_top(const int8_t* d, int count, int radius, int depth, int8_t* top) {
  if (!d || !top || radius < 0 || radius > 63 || depth < 1 || depth > 2 * radius + 1) return -1;
  const int D = 2 * radius + 1;
  int hist[127], idx[127];
  for (int i = 0; i < D; ++i) { hist[i] = 0; idx[i] = i - radius; }
  for (int k = 0; k < count; ++k) {
    int v = (int)d[k] + radius;
    if (v < 0 || v >= D) return -1;
    hist[v]++;
  }
  /* stable descending bubble sort, strict '>' swaps (FastSpacedBMMethod.cl:126-137) */
  int swapped;
  do {
    swapped = 0;
    for (int i = 1; i < D; ++i)
      if (hist[i] > hist[i - 1]) {
        int t = hist[i]; hist[i] = hist[i - 1]; hist[i - 1] = t;
        t = idx[i]; idx[i] = idx[i - 1]; idx[i - 1] = t;
        swapped = 1;
      }
  } while (swapped);
  for (int i = 0; i < depth; ++i) top[i] = (int8_t)idx[i];
  return 0;
}

int oracle_resize_2x_u8(const uint8_t* src, size_t pitch, int w, int h, uint8_t* dst) {
  if (!src || !dst || w < 2 || h < 2) return -1;
  const int dw = 2 * w, dh = 2 * h;
  int* rows = (int*)malloc(sizeof(int) * (size_t)dw * 2);
  if (!rows) return -2;
  for (int dy = 0; dy < dh; ++dy) {
    /* vertical taps: fy = dy/2 - 0.25 */
    /* vertically OpenCV keeps the weights and clips the ROW INDICES (rows -1 and h read rows 0 and h-1) */
    int sy = (dy >> 1) - ((dy & 1) ? 0 : 1);
    const int b0 = (dy & 1) ? 1536 : 512, b1 = 2048 - b0;
    int sy1 = sy + 1;
    if (sy < 0) sy = 0;
    if (sy1 > h - 1) sy1 = h - 1;
    for (int t = 0; t < 2; ++t) {
      const uint8_t* S = src + (size_t)(t ? sy1 : sy) * pitch;
      int* D = rows + t * dw;
      for (int dx = 0; dx < dw; ++dx) {
        int sx = (dx >> 1) - ((dx & 1) ? 0 : 1);
        int a0 = (dx & 1) ? 1536 : 512, a1 = 2048 - a0;
        if (sx < 0) { sx = 0; a0 = 2048; a1 = 0; }
        int sx1 = sx + 1;
        if (sx1 >= w) { sx1 = w - 1; a0 = 2048; a1 = 0; sx = w - 1; }
        D[dx] = S[sx] * a0 + S[sx1] * a1;
      }
    }
    for (int dx = 0; dx < dw; ++dx)
      dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (rows[dx] >> 4)) >> 16) + ((b1 * (rows[dw + dx] >> 4)) >> 16) + 2) >> 2);
  }
  free(rows);
  return 0;
}

int oracle_bm_refine_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, int w, int h, int fullpix_x, int fullpix_y,
                        int passes, int faithful, double* out_xy, int32_t* sads) {
  if (!cur || !prev || !out_xy || passes < 1 || passes > 4) return -1;
  const int W2 = 2 * w, H2 = 2 * h;
  uint8_t* c2 = (uint8_t*)malloc((size_t)W2 * H2);
  uint8_t* p2 = (uint8_t*)malloc((size_t)W2 * H2);
  if (!c2 || !p2) { free(c2); free(p2); return -2; }
  int tx = fullpix_x, ty = fullpix_y, scale = 1, rc = 0;
  for (int i = 1; i <= passes && !rc; ++i) {
    scale *= 2;
    tx *= 2;
    ty *= 2; /* :106-107 */
    if (i == 1) {
      /* :110-111 -- first pass: both are 2x up-samples (of the CURRENT image when faithful) */
      oracle_resize_2x_u8(cur, pitch, w, h, c2);
      if (faithful) memcpy(p2, c2, (size_t)W2 * H2);
      else oracle_resize_2x_u8(prev, pitch, w, h, p2);
    } /* later passes: resize to the same 2x size == copy; faithful: imPrev2x <- imCurr2x (already equal) */
    int spx, spy; /* :113-121 */
    if (tx < 0 && ty < 0) { spx = -tx + 1; spy = -ty + 1; }
    else if (tx < 0 && ty >= 0) { spx = -tx + 1; spy = 1; }
    else if (tx >= 0 && ty < 0) { spx = 1; spy = -ty + 1; }
    else { spx = 1; spy = 1; }
    const int cw = W2 - ((tx < 0 ? -tx : tx) + 2), ch = H2 - ((ty < 0 ? -ty : ty) + 2); /* :123 */
    if (cw <= 0 || ch <= 0) { rc = -3; break; }
    int32_t best = 0;
    int bn = 0, bm = 0, first = 1;
    for (int m = -1; m <= 1; ++m)
      for (int n = -1; n <= 1; ++n) { /* :127-136; stored at (1+n, 1+m): row-major scan == this loop order */
        int64_t acc = 0;
        for (int y = 0; y < ch; ++y) {
          const uint8_t* a = c2 + (size_t)(1 + y) * W2 + 1;
          const uint8_t* b = p2 + (size_t)(spy + m + y) * W2 + spx + n;
          for (int x = 0; x < cw; ++x) acc += a[x] > b[x] ? a[x] - b[x] : b[x] - a[x];
        }
        if (sads) sads[(i - 1) * 9 + (m + 1) * 3 + (n + 1)] = (int32_t)acc;
        if (first || (int32_t)acc < best) { best = (int32_t)acc; bn = n; bm = m; first = 0; }
      }
    tx += bn;
    ty += bm; /* :142 totalOffset + min_loc - (1,1) */
  }
  if (!rc) {
    out_xy[0] = (double)((float)tx / (float)scale); /* :144 */
    out_xy[1] = (double)((float)ty / (float)scale);
  }
  free(c2);
  free(p2);
  return rc;
}
