// Coefficient tables of the frame front-end's resize step, shared by the host implementation (ga3c_host.cpp) and
// the HIP one (ga3c_engine.hip).
//
// The reference resizes with scipy.misc.imresize(gray, [84, 84], 'bilinear') (ga3c/Environment.py:59), i.e.
// Pillow's Image.resize(BILINEAR): a two-pass convolution with a triangle filter whose support grows with the
// downscale factor, 8-bit intermediates and 22-bit fixed-point coefficients.  The tables below are that
// resampler's per-output-pixel (first tap, tap count, integer weights) for the whole-image box; the restatement
// they are checked against is oracle/frame_frontend.py: bilinear_coeffs, which is itself held bit-for-bit to
// Pillow in tests/test_frontend_oracle.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace ga3c {

constexpr int RESAMPLE_PRECISION_BITS = 32 - 8 - 2;

struct ResampleTable {
  int in_size = 0, out_size = 0, ksize = 0;
  std::vector<int32_t> bounds;   // [out_size][2] = first input index, tap count
  std::vector<int32_t> kk;       // [out_size][ksize], unused taps are 0
};

inline ResampleTable make_bilinear_table(int in_size, int out_size) {
#ifdef __clang__
#pragma clang fp contract(off)   // every product and sum rounded separately, as Pillow's C and the oracle do
#endif
  ResampleTable t;
  t.in_size = in_size;
  t.out_size = out_size;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  t.ksize = (int)std::ceil(support) * 2 + 1;
  t.bounds.assign((size_t)out_size * 2, 0);
  t.kk.assign((size_t)out_size * t.ksize, 0);
  const double ss = 1.0 / filterscale;
  std::vector<double> w((size_t)t.ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < t.ksize; ++x) w[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w[x] = a < 1.0 ? 1.0 - a : 0.0;
      ww += w[x];
    }
    if (ww != 0.0)
      for (int x = 0; x < xmax; ++x) w[x] /= ww;
    t.bounds[(size_t)xx * 2] = xmin;
    t.bounds[(size_t)xx * 2 + 1] = xmax;
    for (int x = 0; x < t.ksize; ++x) {
      const double v = w[x] * (double)(1 << RESAMPLE_PRECISION_BITS);
      t.kk[(size_t)xx * t.ksize + x] = w[x] < 0.0 ? (int32_t)(-0.5 + v) : (int32_t)(0.5 + v);
    }
  }
  return t;
}

}  // namespace ga3c
