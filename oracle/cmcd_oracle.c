/*
 * cmcd_oracle.c — plain-C restatement of CMCD's MCD_CAIS_sn / MCD_CAIS_var_sn / MCD_CAIS_UHA_sn bounds.
 * TEST INFRASTRUCTURE ONLY: linked/loaded solely by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  PARITY UNPINNED (the reference is JAX-only, cannot be imported in
 * the build container and ships no tests); pinned by the same PRNG known answers and identities as
 * the NumPy restatement (oracle/cmcd_oracle.py), against which tests/test_oracle_c.py checks it.
 *
 * Written independently of both the NumPy oracle and the HIP kernels: scalar float32 arithmetic,
 * one particle at a time, two network and two gradient evaluations per bridge step exactly as
 * /root/reference/src/mcd_cais.py:46-89 does.  OpenMP over particles (jax.vmap,
 * /root/reference/src/mcdboundingmachine.py:193-203).  The descriptor / layout structs are the ones
 * of include/cmcd_hip.h so that the same params_flat can be handed to both.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "cmcd_hip.h"

#define MAXD 16
#define MAXW 256
static const float HALF_LOG_2PI = 0.91893853320467274178f;

/* ---- jax.random (Threefry-2x32, original layout); /root/reference/src/mcdboundingmachine.py:151-162 */
static inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static void threefry(uint32_t k0, uint32_t k1, uint32_t* x0, uint32_t* x1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t a = *x0 + ks[0], b = *x1 + ks[1];
  for (int g = 1; g <= 5; ++g) {
    for (int r = 0; r < 4; ++r) { a += b; b = rotl(b, R[(g - 1) % 2][r]); b ^= a; }
    a += ks[g % 3];
    b += ks[(g + 1) % 3] + (uint32_t)g;
  }
  *x0 = a; *x1 = b;
}
static void split(const uint32_t key[2], uint32_t first[2], uint32_t second[2]) {
  uint32_t a0 = 0, a1 = 2, b0 = 1, b1 = 3;
  threefry(key[0], key[1], &a0, &a1);
  threefry(key[0], key[1], &b0, &b1);
  first[0] = a0; first[1] = b0; second[0] = a1; second[1] = b1;
}
static float erfinv_giles(float x) {
  float w = -log1pf(-x * x), p;
  if (w < 5.0f) {
    w -= 2.5f;
    p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w;
    p = 0.00021858087f + p * w; p = -0.00125372503f + p * w; p = -0.00417768164f + p * w;
    p = 0.246640727f + p * w; p = 1.50140941f + p * w;
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w;
    p = 0.00573950773f + p * w; p = -0.0076224613f + p * w; p = 0.00943887047f + p * w;
    p = 1.00167406f + p * w; p = 2.83297682f + p * w;
  }
  return p * x;
}
static void normal(const uint32_t key[2], int d, float* out) {
  const int h = (d + 1) / 2;
  const float lo = nextafterf(-1.0f, 0.0f);
  for (int j = 0; j < h; ++j) {
    uint32_t x0 = (uint32_t)j, x1 = (h + j < d) ? (uint32_t)(h + j) : 0u;
    threefry(key[0], key[1], &x0, &x1);
    uint32_t bits[2] = {x0, x1};
    int idx[2] = {j, h + j};
    for (int q = 0; q < 2; ++q) {
      if (idx[q] >= d) continue;
      union { uint32_t u; float f; } c;
      c.u = (bits[q] >> 9) | 0x3F800000u;
      float u = c.f - 1.0f;
      u = u * (1.0f - lo) + lo;
      if (u < lo) u = lo;
      out[idx[q]] = 1.41421356237309504880f * erfinv_giles(u);
    }
  }
}

/* ---- targets: log p and closed-form gradient (/root/reference/src/model_handler.py) */
typedef struct {
  int id, dim, n_mix;
  const float* tc; /* many_gmm: {scale, means[n_mix][2]} */
} Target;

static float gmm_raw(const float* x, float* g) { /* :167-190 */
  static const float mu[3][2] = {{3.0f, 0.0f}, {-2.5f, 0.0f}, {2.0f, 3.0f}};
  static const float P[3][3] = {{1.0f / 0.7f, 0.0f, 20.0f}, {1.0f / 0.7f, 0.0f, 20.0f},
                                {10.256410256410257f, -9.743589743589743f, 10.256410256410257f}};
  static const float lc[3] = {-1.2602857463310935f, -1.2602857463310935f, -1.7725379045882876f};
  float l[3], pd[3][2], m = -INFINITY;
  for (int k = 0; k < 3; ++k) {
    float d0 = x[0] - mu[k][0], d1 = x[1] - mu[k][1];
    pd[k][0] = P[k][0] * d0 + P[k][1] * d1;
    pd[k][1] = P[k][1] * d0 + P[k][2] * d1;
    l[k] = -0.5f * (d0 * pd[k][0] + d1 * pd[k][1]) + lc[k];
    if (l[k] > m) m = l[k];
  }
  float s = 0.f, gx = 0.f, gy = 0.f;
  for (int k = 0; k < 3; ++k) { float e = expf(l[k] - m); s += e; gx += e * pd[k][0]; gy += e * pd[k][1]; }
  g[0] = -gx / s; g[1] = -gy / s;
  return m + logf(s);
}
static float target_eval(const Target* t, const float* z, float* g) {
  const int d = t->dim;
  if (t->id == CMCD_TARGET_GMM) { /* :192-195 */
    float ga[2], gb[2], zf[2] = {z[1], z[0]};
    float fa = gmm_raw(z, ga), fb = gmm_raw(zf, gb);
    float m = fa > fb ? fa : fb, lse = m + logf(expf(fa - m) + expf(fb - m));
    float wa = expf(fa - lse), wb = expf(fb - lse);
    g[0] = wa * ga[0] + wb * gb[1];
    g[1] = wa * ga[1] + wb * gb[0];
    return lse - 0.69314718055994530942f;
  }
  if (t->id == CMCD_TARGET_FUNNEL) { /* :124-143 */
    float v = z[0], ss = 0.f;
    for (int j = 1; j < d; ++j) ss += z[j] * z[j];
    float emv = expf(-v);
    g[0] = -v / 9.0f - 0.5f * (d - 1) + 0.5f * emv * ss;
    for (int j = 1; j < d; ++j) g[j] = -z[j] * emv;
    return -HALF_LOG_2PI - 1.0986122886681098f - v * v / 18.0f - (d - 1) * HALF_LOG_2PI - 0.5f * (d - 1) * v -
           0.5f * emv * ss;
  }
  /* many_gmm :245-281 */
  const float s = t->tc[0];
  const float* mu = t->tc + 1;
  float logit[64], m = -INFINITY;
  const float lc = -2.0f * (logf(s) + HALF_LOG_2PI) - logf((float)t->n_mix);
  for (int k = 0; k < t->n_mix; ++k) {
    float dx = (z[0] - mu[2 * k]) / s, dy = (z[1] - mu[2 * k + 1]) / s;
    logit[k] = -0.5f * (dx * dx + dy * dy) + lc;
    if (logit[k] > m) m = logit[k];
  }
  float sum = 0.f, gx = 0.f, gy = 0.f;
  for (int k = 0; k < t->n_mix; ++k) {
    float e = expf(logit[k] - m);
    sum += e;
    gx += e * (z[0] - mu[2 * k]) / s;
    gy += e * (z[1] - mu[2 * k + 1]) / s;
  }
  float lp = m + logf(sum);
  if (!(lp > -1e4f)) { g[0] = g[1] = 0.f; return -INFINITY; } /* :279-280 */
  g[0] = -gx / sum / s; g[1] = -gy / sum / s;
  return lp;
}

/* ---- score networks */
typedef struct {
  int arch, d, e, in, K;   /* d: output width (= dim); in: geffner width = din + e */
  int din;                 /* state inputs: dim, or 2 dim for concat(z, rho) (MCD_CAIS_UHA_sn, mcdboundingmachine.py:82-98) */
  const float* P;
  const cmcd_layout* lay;
  float* tau; /* dds: [K+1][64] time-path output */
} Net;

static float gelu(float x) { return x * 0.5f * (1.0f + erff(x / sqrtf(2.0f))); }   /* nn_dds.py:167-176 */
static float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); } /* logaddexp(x,0) */

static void dds_time_path(const Net* n, int t, float* tau) { /* nn_dds.py:131-143,155-158 */
  const float* P = n->P;
  float emb[128], h[64];
  for (int j = 0; j < 64; ++j) {
    /* jnp.linspace(0.1, 100, 64) in float32: start (1 - s) + stop s, s = iota / 63, end point exact (nn_dds.py:108) */
    float s = (float)j / 63.0f;
    float c = (j == 63) ? 100.0f : 0.1f * (1.0f - s) + 100.0f * s;
    float arg = c * (float)t + P[n->lay->d_phase + j];
    emb[j] = (float)sin((double)arg);
    emb[64 + j] = (float)cos((double)arg);
  }
  for (int j = 0; j < 64; ++j) {
    float a = P[n->lay->d_tb1 + j];
    for (int k = 0; k < 128; ++k) a += emb[k] * P[n->lay->d_tw1 + k * 64 + j];
    h[j] = gelu(a);
  }
  for (int j = 0; j < 64; ++j) {
    float a = P[n->lay->d_tb2 + j];
    for (int k = 0; k < 64; ++k) a += h[k] * P[n->lay->d_tw2 + k * 64 + j];
    tau[j] = a;
  }
}
static void net_apply(const Net* n, const float* z, int idx, float* out) {
  const float* P = n->P;
  const cmcd_layout* L = n->lay;
  float u[MAXW + 64], v[MAXW + 64];
  if (n->arch == CMCD_ARCH_DDS) { /* nn_dds.py:159-162 */
    const int in = n->din + 64;
    for (int j = 0; j < n->din; ++j) u[j] = z[j];
    memcpy(u + n->din, n->tau + (size_t)idx * 64, 64 * sizeof(float));
    for (int j = 0; j < 64; ++j) {
      float a = P[L->d_sb1 + j];
      for (int k = 0; k < in; ++k) a += u[k] * P[L->d_sw1 + k * 64 + j];
      v[j] = gelu(a);
    }
    for (int j = 0; j < 64; ++j) {
      float a = P[L->d_sb2 + j];
      for (int k = 0; k < 64; ++k) a += v[k] * P[L->d_sw2 + k * 64 + j];
      u[j] = gelu(a);
    }
    for (int j = 0; j < n->d; ++j) {
      float a = P[L->d_sb3 + j];
      for (int k = 0; k < 64; ++k) a += u[k] * P[L->d_sw3 + k * n->d + j];
      out[j] = fminf(fmaxf(a, -1e4f), 1e4f);
    }
    return;
  }
  /* geffner, nn.py:42-72; index clamps like a JAX gather */
  const int in = n->in, ie = idx < n->K ? idx : n->K - 1;
  for (int j = 0; j < n->din; ++j) u[j] = z[j];
  for (int j = 0; j < n->e; ++j) u[n->din + j] = P[L->g_emb + (size_t)ie * n->e + j];
  const int64_t W[2] = {L->g_w1, L->g_w2}, B[2] = {L->g_b1, L->g_b2};
  for (int l = 0; l < 2; ++l) {
    for (int j = 0; j < in; ++j) {
      float a = P[B[l] + j];
      for (int k = 0; k < in; ++k) a += u[k] * P[W[l] + (size_t)k * in + j];
      v[j] = u[j] + softplus(a);
    }
    memcpy(u, v, in * sizeof(float));
  }
  const float f = P[L->g_factor];
  for (int j = 0; j < n->d; ++j) {
    float a = P[L->g_b3 + j];
    for (int k = 0; k < in; ++k) a += u[k] * P[L->g_w3 + (size_t)k * n->d + j];
    out[j] = a * f;
  }
}

static float log_prob_kernel(const float* x, const float* mean, float scale, int d) { /* mcd_utils.py:19-21 */
  float s = 0.f;
  for (int j = 0; j < d; ++j) {
    float df = x[j] - mean[j];
    s += -(df * df) / (2.0f * scale * scale) - logf(scale) - HALF_LOG_2PI;
  }
  return s;
}

int cmcd_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void cmcd_oracle_set_threads(int n) {   /* bench.py's one-thread line */
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* Host pointers everywhere.  Returns 0, or -2 for unsupported (lgcp, widths > 256). */
int cmcd_oracle_bound(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                      const float* P, const float* target_consts, int64_t n_target, float* out_loss,
                      float* out_z) {
  const int d = desc->dim, K = desc->nbridges;
  if (desc->target == CMCD_TARGET_LGCP || d > MAXD) return -2;
  const int uha = desc->mode == CMCD_MODE_CAIS_UHA_SN;
  if (desc->mode != CMCD_MODE_CAIS_SN && desc->mode != CMCD_MODE_CAIS_VAR_SN && !uha) return -2;
  const int din = uha ? 2 * d : d;
  Net net = {desc->arch, d, desc->emb_dim, din + desc->emb_dim, K, din, P, lay, NULL};
  if (desc->arch == CMCD_ARCH_GEFFNER && net.in > MAXW) return -2;
  Target tgt = {desc->target, d, desc->target == CMCD_TARGET_MANY_GMM ? (int)((n_target - 1) / 2) : 0, target_consts};

  /* betas (mcdboundingmachine.py:146-149) and eps schedule (mcd_cais.py:34-44,54-59) */
  float* beta = (float*)malloc(sizeof(float) * K);
  float* epsv = (float*)malloc(sizeof(float) * K);
  {
    const int G = desc->ngrid;
    float gy[40], tot = 0.f, run = 0.f;
    for (int i = 0; i <= G; ++i) tot += P[lay->mgridref_y + i];
    gy[0] = 0.f;
    for (int i = 0; i <= G; ++i) { run += P[lay->mgridref_y + i]; gy[i + 1] = run / tot; }
    const float eps0 = P[lay->eps];
    for (int i = 0; i < K; ++i) {
      float x = (float)(i + 1) / (float)(K + 1);
      int j = 1;
      while (j < G + 1 && (float)j / (float)(G + 1) <= x) ++j;
      float x0 = (float)(j - 1) / (float)(G + 1), x1 = (float)j / (float)(G + 1);
      beta[i] = gy[j - 1] + (x - x0) / (x1 - x0) * (gy[j] - gy[j - 1]);
      if (desc->eps_schedule == CMCD_EPS_COS_SQ || uha) {   /* the 2nd-order mode's body fixes cos^2 (mcd_under_lp_a_cais.py:33-48) */
        float c = cosf(((float)i / (float)K + 0.008f) / 1.008f * 0.5f * 3.14159265358979323846f);
        epsv[i] = eps0 * c * c;
      } else if (desc->eps_schedule == CMCD_EPS_LINEAR) {
        epsv[i] = (0.0001f - eps0) / (float)(K - 1) * (float)i + eps0;
      } else {
        epsv[i] = eps0;
      }
    }
  }
  if (desc->arch == CMCD_ARCH_DDS) { /* the time path does not depend on the particle: once per index */
    net.tau = (float*)malloc(sizeof(float) * 64 * (K + 1));
    for (int t = 0; t <= K; ++t) dds_time_path(&net, t, net.tau + (size_t)t * 64);
  }
  const int var_mode = desc->mode == CMCD_MODE_CAIS_VAR_SN;
  const float clip = var_mode ? 1e2f : 1e3f;

#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t p = 0; p < n; ++p) {
    float mean[MAXD], sd[MAXD], z[MAXD], zn[MAXD], noise[MAXD], gp[MAXD], s[MAXD], fk[MAXD], bk[MAXD];
    for (int j = 0; j < d; ++j) { mean[j] = P[lay->vd_mean + j]; sd[j] = expf(P[lay->vd_logdiag + j]); }
    uint32_t key[2] = {0u, (uint32_t)seeds[p]}, a[2], b[2], c[2], tmp[2], gen[2];
    split(key, a, b);
    normal(a, d, noise);
    float w = 0.f;
    for (int j = 0; j < d; ++j) { /* diag_gauss.py:26-62 */
      z[j] = sd[j] * noise[j] + mean[j];
      float dz = z[j] - mean[j];
      w -= -(dz * dz) / (2.0f * sd[j] * sd[j]) - logf(sd[j]) - HALF_LOG_2PI;
    }
    split(b, c, tmp);
    if (uha) {
      /* MCD_CAIS_UHA_sn: /root/reference/src/mcd_under_lp_a_cais.py:42-112 — state (z, rho), both network calls on
       * concat(z, rho) / concat(z, rho') at index i, eta_aux = gamma eps, one leap-frog step, clip 1e2 on grad log p only */
      float rho[MAXD], rhop[MAXD], rpp[MAXD], x[2 * MAXD], mf[MAXD], mb[MAXD], zero[MAXD], uf[MAXD];
      uint32_t rk[2], gp2[2];
      const float gamma = P[lay->gamma];
      for (int j = 0; j < d; ++j) zero[j] = 0.f;
      split(c, rk, gp2);                    /* :92 */
      normal(rk, d, rho);                   /* :93 */
      w -= log_prob_kernel(rho, zero, 1.0f, d);   /* :96-97 */
      split(gp2, tmp, gen);                 /* :100 */
      for (int i = 0; i < K; ++i) {
        const float be = beta[i], eps = epsv[i], eta = gamma * eps, scale = sqrtf(2.0f * eta);
        target_eval(&tgt, z, gp);
        for (int j = 0; j < d; ++j) {
          float gq = -(z[j] - mean[j]) / (sd[j] * sd[j]), g = fminf(fmaxf(gp[j], -1e2f), 1e2f);
          uf[j] = -1.0f * (be * g + (1.0f - be) * gq);                  /* :23-30,46 */
          x[j] = z[j]; x[d + j] = rho[j];
        }
        net_apply(&net, x, i, s);
        for (int j = 0; j < d; ++j) mf[j] = rho[j] * (1.0f - eta) - 2.0f * eta * s[j];   /* :52-54 */
        uint32_t gk[2], hk[2];
        split(gen, gk, hk);                 /* :55 */
        normal(gk, d, noise);
        for (int j = 0; j < d; ++j) {
          rhop[j] = mf[j] + scale * noise[j];                           /* :58-59 */
          rpp[j] = rhop[j] - eps * uf[j] / 2.0f;                        /* :62 */
          zn[j] = z[j] + eps * rpp[j];                                  /* :63 */
          x[d + j] = rhop[j];                                           /* :77: old z, new momentum */
        }
        net_apply(&net, x, i, s);
        for (int j = 0; j < d; ++j) mb[j] = rhop[j] * (1.0f - eta) + 2.0f * eta * s[j];  /* :78-80 */
        w += log_prob_kernel(rho, mb, scale, d) - log_prob_kernel(rhop, mf, scale, d);   /* :83-88 */
        target_eval(&tgt, zn, gp);
        for (int j = 0; j < d; ++j) {
          float gq = -(zn[j] - mean[j]) / (sd[j] * sd[j]), g = fminf(fmaxf(gp[j], -1e2f), 1e2f);
          float ub = -1.0f * (be * g + (1.0f - be) * gq);               /* :65 */
          rho[j] = rpp[j] - eps * ub / 2.0f;                            /* :67 */
        }
        split(hk, tmp, gen);                /* :84 */
        memcpy(z, zn, sizeof(float) * d);
      }
      w += log_prob_kernel(rho, zero, 1.0f, d);   /* :112 */
      w += target_eval(&tgt, z, gp);
      out_loss[p] = -w;
      memcpy(out_z + p * d, z, sizeof(float) * d);
      continue;
    }
    split(c, tmp, gen); /* mcd_cais.py:94 */
    for (int i = 0; i < K; ++i) {
      const float be = beta[i], eps = epsv[i], scale = sqrtf(2.0f * eps);
      /* forward kernel, mcd_cais.py:52-67 */
      target_eval(&tgt, z, gp);
      net_apply(&net, z, i, s);
      for (int j = 0; j < d; ++j) {
        float gq = -(z[j] - mean[j]) / (sd[j] * sd[j]), g = gp[j];
        if (desc->grad_clipping) { g = fminf(fmaxf(g, -clip), clip); if (var_mode) gq = fminf(fmaxf(gq, -clip), clip); }
        float uf = -1.0f * (be * g + (1.0f - be) * gq);
        fk[j] = z[j] - eps * uf - eps * s[j];
      }
      uint32_t gk[2], hk[2];
      split(gen, gk, hk);
      normal(gk, d, noise);
      for (int j = 0; j < d; ++j) zn[j] = fk[j] + scale * noise[j];
      /* backward kernel, mcd_cais.py:71-79 */
      target_eval(&tgt, zn, gp);
      net_apply(&net, zn, i + 1, s);
      for (int j = 0; j < d; ++j) {
        float gq = -(zn[j] - mean[j]) / (sd[j] * sd[j]), g = gp[j];
        if (desc->grad_clipping) { g = fminf(fmaxf(g, -clip), clip); if (var_mode) gq = fminf(fmaxf(gq, -clip), clip); }
        float ub = -1.0f * (be * g + (1.0f - be) * gq);
        bk[j] = zn[j] - eps * ub + eps * s[j];
      }
      w += log_prob_kernel(z, bk, scale, d) - log_prob_kernel(zn, fk, scale, d); /* :82-86 */
      split(hk, tmp, gen);                                                       /* :87 */
      memcpy(z, zn, sizeof(float) * d);
    }
    w += target_eval(&tgt, z, gp); /* mcdboundingmachine.py:178 */
    out_loss[p] = -w;
    memcpy(out_z + p * d, z, sizeof(float) * d);
  }
  free(beta); free(epsv); free(net.tau);
  return 0;
}
