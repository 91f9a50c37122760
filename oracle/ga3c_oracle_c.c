/* ga3c_oracle_c.c -- f32 C port of the oracle (TEST INFRASTRUCTURE ONLY; never linked into the product).
 *
 * Used (a) as a second, independent restatement checked against oracle/ga3c_oracle.py, and (b) as the
 * `cpu_baseline` leg of bench.py ("kind": "port"): the reference's TensorFlow-1.x CPU path cannot run
 * here (TensorFlow absent), so this multithreaded port of the same graph is what gets timed on the GPU
 * box's host cores.  NN parity is unpinned at the TensorFlow boundary (see ga3c_oracle.py header).
 *
 * Follows /root/reference/ga3c: conv NetworkVP.py:212-228 (HWIO, SAME, ReLU) wired as NetworkDNav.py:81-90;
 * heads + loss NetworkVP_discrate.py:60-85 (default branch, MIN_POLICY supported); RMSProp :99-105,130
 * (TF ApplyRMSProp arithmetic, ms slot initialised to ones by the caller).
 *
 * Arena layout = TF variable order: conv11/w,b conv12/w,b dense1/w,b logits_v/w,b logits_p/w,b.
 * Build: gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IMG 84
#define XS (84 * 84 * 4)
#define O1 21
#define N1S (21 * 21 * 16)
#define O2 11
#define FLAT (11 * 11 * 32)
#define HID 256
#define OFF_W1 0
#define OFF_B1 4096
#define OFF_W2 4112
#define OFF_B2 12304
#define OFF_WD 12336
#define OFF_BD (12336 + FLAT * HID)
#define OFF_WV (OFF_BD + HID)
#define OFF_BV (OFF_WV + HID)
#define OFF_WP (OFF_BV + 1)
#define SB 8 /* samples per dense block */

int64_t ga3c_oc_param_count(int A) { return OFF_WP + (int64_t)HID * A + A; }
int ga3c_oc_max_threads(void) { return omp_get_max_threads(); }
void ga3c_oc_set_threads(int n) { omp_set_num_threads(n); }

static void conv1_fwd(const float* x, const float* w, const float* b, float* n1) {
  for (int i = 0; i < O1; ++i)
    for (int j = 0; j < O1; ++j) {
      float acc[16];
      for (int o = 0; o < 16; ++o) acc[o] = b[o];
      for (int u = 0; u < 8; ++u) {
        const int yy = 4 * i - 2 + u;
        if (yy < 0 || yy >= IMG) continue;
        for (int v = 0; v < 8; ++v) {
          const int xx = 4 * j - 2 + v;
          if (xx < 0 || xx >= IMG) continue;
          const float* px = x + (yy * IMG + xx) * 4;
          const float* pw = w + (u * 8 + v) * 4 * 16;
          for (int c = 0; c < 4; ++c) {
            const float xv = px[c];
            for (int o = 0; o < 16; ++o) acc[o] += xv * pw[c * 16 + o];
          }
        }
      }
      float* out = n1 + (i * O1 + j) * 16;
      for (int o = 0; o < 16; ++o) out[o] = acc[o] > 0.f ? acc[o] : 0.f;
    }
}

static void conv2_fwd(const float* n1, const float* w, const float* b, float* n2) {
  for (int i = 0; i < O2; ++i)
    for (int j = 0; j < O2; ++j) {
      float acc[32];
      for (int o = 0; o < 32; ++o) acc[o] = b[o];
      for (int u = 0; u < 4; ++u) {
        const int yy = 2 * i - 1 + u;
        if (yy < 0 || yy >= O1) continue;
        for (int v = 0; v < 4; ++v) {
          const int xx = 2 * j - 1 + v;
          if (xx < 0 || xx >= O1) continue;
          const float* px = n1 + (yy * O1 + xx) * 16;
          const float* pw = w + (u * 4 + v) * 16 * 32;
          for (int c = 0; c < 16; ++c) {
            const float xv = px[c];
            for (int o = 0; o < 32; ++o) acc[o] += xv * pw[c * 32 + o];
          }
        }
      }
      float* out = n2 + (i * O2 + j) * 32;
      for (int o = 0; o < 32; ++o) out[o] = acc[o] > 0.f ? acc[o] : 0.f;
    }
}

/* d1[s][:] = relu(flat[s][:] Wd + bd) for a block of up to SB samples (weight row loaded once per block) */
static void dense_fwd_block(const float* flat, int ns, const float* wd, const float* bd, float* d1) {
  float acc[SB][HID];
  for (int s = 0; s < ns; ++s) memcpy(acc[s], bd, HID * sizeof(float));
  for (int k = 0; k < FLAT; ++k) {
    const float* wr = wd + (size_t)k * HID;
    for (int s = 0; s < ns; ++s) {
      const float f = flat[(size_t)s * FLAT + k];
      if (f == 0.f) continue;
      for (int n = 0; n < HID; ++n) acc[s][n] += f * wr[n];
    }
  }
  for (int s = 0; s < ns; ++s)
    for (int n = 0; n < HID; ++n) d1[(size_t)s * HID + n] = acc[s][n] > 0.f ? acc[s][n] : 0.f;
}

static void heads_fwd(const float* d1, const float* th, int A, float min_policy, float* z, float* p, float* v) {
  const float *wv = th + OFF_WV, *wp = th + OFF_WP, *bp = th + OFF_WP + HID * A;
  float vv = th[OFF_BV];
  for (int k = 0; k < HID; ++k) vv += d1[k] * wv[k];
  *v = vv;
  for (int o = 0; o < A; ++o) z[o] = bp[o];
  for (int k = 0; k < HID; ++k)
    for (int o = 0; o < A; ++o) z[o] += d1[k] * wp[k * A + o];
  float zmax = z[0];
  for (int o = 1; o < A; ++o) zmax = z[o] > zmax ? z[o] : zmax;
  float sum = 0.f;
  for (int o = 0; o < A; ++o) { p[o] = expf(z[o] - zmax); sum += p[o]; }
  for (int o = 0; o < A; ++o) p[o] = (p[o] / sum + min_policy) / (1.0f + min_policy * A);
}

/* forward for B samples; work buffers n1,n2,d1 sized for B; z may be NULL */
int ga3c_oc_forward(const float* th, int A, const float* x, int B, float min_policy, float* n1, float* n2,
                    float* d1, float* z, float* p, float* v) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int b0 = 0; b0 < B; b0 += SB) {
    const int ns = B - b0 < SB ? B - b0 : SB;
    for (int s = 0; s < ns; ++s) {
      const int b = b0 + s;
      conv1_fwd(x + (size_t)b * XS, th + OFF_W1, th + OFF_B1, n1 + (size_t)b * N1S);
      conv2_fwd(n1 + (size_t)b * N1S, th + OFF_W2, th + OFF_B2, n2 + (size_t)b * FLAT);
    }
    dense_fwd_block(n2 + (size_t)b0 * FLAT, ns, th + OFF_WD, th + OFF_BD, d1 + (size_t)b0 * HID);
    for (int s = 0; s < ns; ++s) {
      const int b = b0 + s;
      float zz[64];
      heads_fwd(d1 + (size_t)b * HID, th, A, min_policy, zz, p + (size_t)b * A, v + b);
      if (z) memcpy(z + (size_t)b * A, zz, A * sizeof(float));
    }
  }
  return 0;
}

/* Convenience wrapper owning its work buffers (what bench.py times for predictions/s). */
int ga3c_oc_predict(const float* th, int A, const float* x, int B, float* p, float* v) {
  float* n1 = (float*)malloc((size_t)B * N1S * sizeof(float));
  float* n2 = (float*)malloc((size_t)B * FLAT * sizeof(float));
  float* d1 = (float*)malloc((size_t)B * HID * sizeof(float));
  if (!n1 || !n2 || !d1) { free(n1); free(n2); free(d1); return -1; }
  ga3c_oc_forward(th, A, x, B, 0.f, n1, n2, d1, NULL, p, v);
  free(n1); free(n2); free(d1);
  return 0;
}

/* One full training step: gradients of the sum-over-batch A3C loss (default branch) into grad[],
 * then (if lr >= 0) TF-1.x RMSProp into th[] / ms[].  losses[3] = cost_p_1_agg, cost_p_2_agg, cost_v. */
int ga3c_oc_train(float* th, float* ms, float* grad, int A, const float* x, const float* y_r, const float* act, int B,
                  float lr, float beta, float log_eps, float min_policy, float rho, float eps, float* losses) {
  const int64_t n = ga3c_oc_param_count(A);
  float* n1 = (float*)malloc((size_t)B * N1S * sizeof(float));
  float* n2 = (float*)malloc((size_t)B * FLAT * sizeof(float));
  float* d1 = (float*)malloc((size_t)B * HID * sizeof(float));
  float* z = (float*)malloc((size_t)B * A * sizeof(float));
  float* p = (float*)malloc((size_t)B * A * sizeof(float));
  float* v = (float*)malloc((size_t)B * sizeof(float));
  float* dz = (float*)malloc((size_t)B * A * sizeof(float));
  float* dv = (float*)malloc((size_t)B * sizeof(float));
  float* dd1 = (float*)malloc((size_t)B * HID * sizeof(float));
  float* dn2 = (float*)malloc((size_t)B * FLAT * sizeof(float));
  float* dn1 = (float*)malloc((size_t)B * N1S * sizeof(float));
  const int T = omp_get_max_threads();
  const int small = 4096 + 16 + 8192 + 32;   /* conv grads: thread-private, reduced afterwards */
  float* priv = (float*)calloc((size_t)T * small, sizeof(float));
  if (!n1 || !n2 || !d1 || !z || !p || !v || !dz || !dv || !dd1 || !dn2 || !dn1 || !priv) return -1;
  ga3c_oc_forward(th, A, x, B, min_policy, n1, n2, d1, z, p, v);
  memset(grad, 0, (size_t)n * sizeof(float));

  /* loss + head gradients, serial: tiny */
  double c1 = 0, c2 = 0, cv = 0;
  const float denom = 1.0f + min_policy * A;
  for (int b = 0; b < B; ++b) {
    const float* pb = p + (size_t)b * A;
    const float* ab = act + (size_t)b * A;
    const float adv = y_r[b] - v[b];
    float sel = 0.f, ent = 0.f, s[64], gs[64], dot = 0.f;
    for (int o = 0; o < A; ++o) sel += pb[o] * ab[o];
    const float gsel = sel >= log_eps ? 1.0f / sel : 0.f;
    for (int o = 0; o < A; ++o) {
      const float lp = logf(pb[o] > log_eps ? pb[o] : log_eps);
      ent += lp * pb[o];
      s[o] = pb[o] * denom - min_policy;
      gs[o] = (-(adv * gsel) * ab[o] + beta * (lp + (pb[o] >= log_eps ? 1.0f : 0.f))) / denom;
      dot += gs[o] * s[o];
    }
    for (int o = 0; o < A; ++o) dz[(size_t)b * A + o] = s[o] * (gs[o] - dot);
    dv[b] = v[b] - y_r[b];
    c1 += logf(sel > log_eps ? sel : log_eps) * adv;
    c2 += -beta * ent;
    cv += 0.5 * (double)(y_r[b] - v[b]) * (y_r[b] - v[b]);
  }
  if (losses) { losses[0] = (float)c1; losses[1] = (float)c2; losses[2] = (float)cv; }

  /* head weight grads + dd1 */
  {
    float *gwv = grad + OFF_WV, *gwp = grad + OFF_WP, *gbp = grad + OFF_WP + HID * A;
    const float *wv = th + OFF_WV, *wp = th + OFF_WP;
    for (int b = 0; b < B; ++b) {
      grad[OFF_BV] += dv[b];
      for (int o = 0; o < A; ++o) gbp[o] += dz[(size_t)b * A + o];
      for (int k = 0; k < HID; ++k) {
        const float h = d1[(size_t)b * HID + k];
        gwv[k] += h * dv[b];
        float acc = dv[b] * wv[k];
        for (int o = 0; o < A; ++o) {
          gwp[k * A + o] += h * dz[(size_t)b * A + o];
          acc += dz[(size_t)b * A + o] * wp[k * A + o];
        }
        dd1[(size_t)b * HID + k] = h > 0.f ? acc : 0.f;
      }
    }
  }
  /* dense1: bias, weight (parallel over rows of Wd), input gradient (parallel over samples) */
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < HID; ++k) grad[OFF_BD + k] += dd1[(size_t)b * HID + k];
#pragma omp parallel for schedule(static)
  for (int k = 0; k < FLAT; ++k) {
    float* gr = grad + OFF_WD + (size_t)k * HID;
    for (int b = 0; b < B; ++b) {
      const float f = n2[(size_t)b * FLAT + k];
      if (f == 0.f) continue;
      const float* dr = dd1 + (size_t)b * HID;
      for (int m = 0; m < HID; ++m) gr[m] += f * dr[m];
    }
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (int b0 = 0; b0 < B; b0 += SB) {
    const int ns = B - b0 < SB ? B - b0 : SB;
    for (int k = 0; k < FLAT; ++k) {
      const float* wr = th + OFF_WD + (size_t)k * HID;
      for (int s = 0; s < ns; ++s) {
        const size_t idx = (size_t)(b0 + s) * FLAT + k;
        if (!(n2[idx] > 0.f)) { dn2[idx] = 0.f; continue; }
        const float* dr = dd1 + (size_t)(b0 + s) * HID;
        float acc = 0.f;
        for (int m = 0; m < HID; ++m) acc += dr[m] * wr[m];
        dn2[idx] = acc;
      }
    }
  }
  /* conv2 backward (dW2, db2, dn1) and conv1 backward (dW1, db1): per sample, thread-private accumulators */
#pragma omp parallel
  {
    float* mine = priv + (size_t)omp_get_thread_num() * small;
    float *gw1 = mine, *gb1 = mine + 4096, *gw2 = mine + 4112, *gb2 = mine + 4112 + 8192;
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      const float* n1b = n1 + (size_t)b * N1S;
      float* dn1b = dn1 + (size_t)b * N1S;
      memset(dn1b, 0, N1S * sizeof(float));
      for (int i = 0; i < O2; ++i)
        for (int j = 0; j < O2; ++j) {
          const float* dy = dn2 + (size_t)b * FLAT + (i * O2 + j) * 32;
          for (int o = 0; o < 32; ++o) gb2[o] += dy[o];
          for (int u = 0; u < 4; ++u) {
            const int yy = 2 * i - 1 + u;
            if (yy < 0 || yy >= O1) continue;
            for (int vv = 0; vv < 4; ++vv) {
              const int xx = 2 * j - 1 + vv;
              if (xx < 0 || xx >= O1) continue;
              const float* px = n1b + (yy * O1 + xx) * 16;
              float* pdx = dn1b + (yy * O1 + xx) * 16;
              const float* pw = th + OFF_W2 + (u * 4 + vv) * 16 * 32;
              float* pg = gw2 + (u * 4 + vv) * 16 * 32;
              for (int c = 0; c < 16; ++c) {
                const float xv = px[c];
                float acc = 0.f;
                for (int o = 0; o < 32; ++o) {
                  pg[c * 32 + o] += xv * dy[o];
                  acc += dy[o] * pw[c * 32 + o];
                }
                pdx[c] += acc;
              }
            }
          }
        }
      for (int e = 0; e < N1S; ++e)
        if (!(n1b[e] > 0.f)) dn1b[e] = 0.f;
      const float* xb = x + (size_t)b * XS;
      for (int i = 0; i < O1; ++i)
        for (int j = 0; j < O1; ++j) {
          const float* dy = dn1b + (i * O1 + j) * 16;
          for (int o = 0; o < 16; ++o) gb1[o] += dy[o];
          for (int u = 0; u < 8; ++u) {
            const int yy = 4 * i - 2 + u;
            if (yy < 0 || yy >= IMG) continue;
            for (int vv = 0; vv < 8; ++vv) {
              const int xx = 4 * j - 2 + vv;
              if (xx < 0 || xx >= IMG) continue;
              const float* px = xb + (yy * IMG + xx) * 4;
              float* pg = gw1 + (u * 8 + vv) * 4 * 16;
              for (int c = 0; c < 4; ++c) {
                const float xv = px[c];
                for (int o = 0; o < 16; ++o) pg[c * 16 + o] += xv * dy[o];
              }
            }
          }
        }
    }
  }
  for (int t = 0; t < T; ++t) {
    const float* mine = priv + (size_t)t * small;
    for (int e = 0; e < 4096 + 16; ++e) grad[OFF_W1 + e] += mine[e];
    for (int e = 0; e < 8192 + 32; ++e) grad[OFF_W2 + e] += mine[4112 + e];
  }
  if (lr >= 0.f) {
    const float omr = 1.0f - rho;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const float g = grad[i];
      float m = ms[i];
      m += (g * g - m) * omr;
      ms[i] = m;
      th[i] -= (g * lr) / sqrtf(eps + m);
    }
  }
  free(n1); free(n2); free(d1); free(z); free(p); free(v); free(dz); free(dv); free(dd1); free(dn2); free(dn1);
  free(priv);
  return 0;
}
