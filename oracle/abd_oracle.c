/*
 * abd_oracle.c -- plain-C CPU restatement of the abdpymc joint logp + gradient.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * file.  The product package abdpymc_amd never links or loads it (it fails loudly without its HIP
 * library instead).
 *
 * It follows the reference's algorithm column by column (one individual at a time), in the
 * recurrence form the reference itself provides and tests as equivalent to its dense design:
 *   constrain_infections      abdpymc/abd.py:640-667, 732-771, 792-818, 560-601
 *   temp response (scan form) abdpymc/abd.py:277-293  (== dense abd.py:242-274, test_abd.py:987-1011)
 *   perm response             abdpymc/abd.py:296-306
 *   model_n/s_response        abdpymc/abd.py:309-393  (S boosts are exactly 1: temp unused, abd.py:272)
 *   logistic + Normal lik     abdpymc/abd.py:445-469, 556-557
 *   priors + Jacobians        PyMC v5 closed forms (pymc is an unpinned dependency, pyproject.toml:10)
 *
 * Pinning: checked against oracle/abd_oracle.py (which is pinned on the reference's known-answer tests)
 * in tests/test_oracle_c.py.  Joint logp/dlogp: parity unpinned against the reference itself (no
 * reference test evaluates them).
 *
 * OpenMP over individuals / observations; it doubles as the "strong CPU" baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NT 17
static const double LOG_2PI = 1.8378770664093453;

static double sigmoid(double t) { return 1.0 / (1.0 + exp(-t)); }
static double softplus(double t) { return (t > 0 ? t : 0) + log1p(exp(-fabs(t))); }

/* Closed-form priors + transform Jacobians and their gradient. */
static double priors(const double* t, int G, double cells, double n1, double N, double m1, double* g) {
  double lp = 0.0;
  for (int k = 0; k < NT; ++k) g[k] = 0.0;
  {
    double L0 = -softplus(-t[0]), L1 = -softplus(t[0]), p = sigmoid(t[0]);
    double bm1 = (double)(G - 1) - 1.0;
    double lnB = lgamma(1.0) + lgamma((double)(G - 1)) - lgamma((double)G);
    lp += (bm1 == 0.0 ? 0.0 : bm1 * L1) - lnB + L0 + L1 + n1 * L0 + (cells - n1) * L1;
    g[0] = (1.0 + n1) * (1.0 - p) - p * (bm1 + 1.0 + (cells - n1));
  }
  {
    const int gk[5] = {1, 2, 5, 8, 9};
    const double mu[5] = {2.0, 1.0, 2.0, 1.0, 1.0};
    for (int q = 0; q < 5; ++q) {
      double al = mu[q] * mu[q] / 0.25, be = mu[q] / 0.25, x = exp(t[gk[q]]);
      lp += al * log(be) - lgamma(al) + al * t[gk[q]] - be * x;
      g[gk[q]] = al - be * x;
    }
  }
  {
    const int bk[2] = {3, 6};
    for (int q = 0; q < 2; ++q) {
      int k = bk[q];
      double L0 = -softplus(-t[k]), L1 = -softplus(t[k]), r = sigmoid(t[k]);
      double lnB = lgamma(10.0) + lgamma(1.0) - lgamma(11.0);
      lp += 9.0 * L0 - lnB + L0 + L1;
      g[k] = 10.0 * (1.0 - r) - r;
    }
  }
  {
    double L0 = -softplus(-t[7]), L1 = -softplus(t[7]), q = sigmoid(t[7]);
    lp += L0 + L1 + m1 * L0 + (N - m1) * L1;
    g[7] = (1.0 + m1) * (1.0 - q) - q * (1.0 + (N - m1));
  }
  {
    const int nk[6] = {4, 10, 11, 12, 14, 15};
    const double mu[6] = {-2.0, -2.0, -1.0, 2.0, -1.0, 2.0};
    const double sd[6] = {1.0, 1.0, 0.5, 0.5, 0.5, 0.5};
    for (int q = 0; q < 6; ++q) {
      double z = (t[nk[q]] - mu[q]) / sd[q];
      lp += -0.5 * z * z - log(sd[q]) - 0.5 * LOG_2PI;
      g[nk[q]] = -z / sd[q];
    }
  }
  {
    const int ek[2] = {13, 16};
    for (int q = 0; q < 2; ++q) {
      double x = exp(t[ek[q]]);
      lp += -x + t[ek[q]];
      g[ek[q]] = -x + 1.0;
    }
  }
  return lp;
}

/* One individual's column: i = constrain(i_raw, pcrpos, splits).  All arrays length G. */
static void constrain_column(int G, int n_splits, const int* splits, const int8_t* raw, const int8_t* pcr,
                             int8_t* out) {
  int8_t tmp[1024];
  if (n_splits == 0) {
    for (int g = 0; g < G; ++g) tmp[g] = (raw[g] + (pcr ? pcr[g] : 0)) > 0; /* abd.py:643-647 */
  } else {
    int edges[4];
    edges[0] = 0;
    for (int k = 0; k < n_splits; ++k) edges[k + 1] = splits[k];
    edges[n_splits + 1] = G;
    for (int c = 0; c <= n_splits; ++c) {
      int lo = edges[c], hi = edges[c + 1];
      int any = 0, cum = 0;
      for (int g = lo; g < hi; ++g) any |= pcr ? pcr[g] != 0 : 0; /* pcrpos.any(axis=0)   abd.py:771 */
      for (int g = lo; g < hi; ++g) {
        cum += raw[g];
        int8_t m = cum > 1 ? 0 : raw[g]; /* where(cumsum > 1, 0, arr)   abd.py:818 */
        tmp[g] = any ? pcr[g] : m;
      }
    }
  }
  /* scan with taps -3,-2,-1 on its own output   abd.py:560-601 */
  for (int g = 0; g < G; ++g) {
    int b = (g >= 1 && out[g - 1]) || (g >= 2 && out[g - 2]) || (g >= 3 && out[g - 3]);
    out[g] = b ? 0 : tmp[g];
  }
}

typedef struct {
  double ll, gh, ghc, ghu, ghd, ghx, gws;
} lik_acc;

/*
 * logp + gradient.  Layouts as the reference holds them: vacs/pcrpos (N, G); i_raw (G, N); waner (N).
 * Observation lists in any order.  work: caller-provided scratch of 6*G*N doubles + G*N bytes, or NULL.
 * Returns 0, or -1 on bad sizes.
 */
int abd_oracle_logp_dlogp(int G, int N, int n_splits, const int* splits, const int8_t* vacs, const int8_t* pcrpos,
                          int64_t K_s, const int32_t* s_gap, const int32_t* s_ind, const double* s_x, const double* s_y,
                          int64_t K_n, const int32_t* n_gap, const int32_t* n_ind, const double* n_x, const double* n_y,
                          const int8_t* i_raw, const int8_t* waner, const double* theta, double* logp, double* grad,
                          int8_t* i_out /* (G,N) or NULL */, int nthreads) {
  if (G < 2 || G > 1024 || N < 1 || n_splits < 0 || n_splits > 2) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  const double perm_n = exp(theta[1]), temp_n = exp(theta[2]), rho_n = sigmoid(theta[3]), init_n = theta[4];
  const double perm_s = exp(theta[5]), rho_s = sigmoid(theta[6]), init_s = theta[10];
  const double b_n = theta[11], d_n = theta[12], sig_n = exp(theta[13]);
  const double b_s = theta[14], d_s = theta[15], sig_s = exp(theta[16]);
  const size_t cells = (size_t)G * N;
  /* per-cell state, individual-major [j*G + g] */
  double* mu_n = (double*)malloc(cells * sizeof(double));
  double* mu_s = (double*)malloc(cells * sizeof(double));
  double* un = (double*)malloc(cells * sizeof(double));
  double* dn = (double*)malloc(cells * sizeof(double));
  double* ds = (double*)malloc(cells * sizeof(double));
  int8_t* cums = (int8_t*)malloc(cells * 2);
  if (!mu_n || !mu_s || !un || !dn || !ds || !cums) return -1;
  int8_t* cum_n = cums;
  int8_t* cum_s = cums + cells;
  long n1 = 0, m1 = 0;

#pragma omp parallel for schedule(static) reduction(+ : n1, m1)
  for (int j = 0; j < N; ++j) {
    int8_t raw[1024], inf[1024];
    for (int g = 0; g < G; ++g) {
      raw[g] = i_raw[(size_t)g * N + j];
      n1 += raw[g];
    }
    m1 += waner[j];
    constrain_column(G, n_splits, splits, raw, pcrpos ? pcrpos + (size_t)j * G : 0, inf);
    const double rj = rho_s * waner[j] + 1 - waner[j]; /* abd.py:374 */
    double tn = 0, dtn = 0, ts = 0, dts = 0;
    int ci = 0, civ = 0;
    for (int g = 0; g < G; ++g) {
      const size_t o = (size_t)j * G + g;
      const double e_i = inf[g], e_v = vacs[o];
      dtn = rho_n * dtn + tn;       /* d/drho of prev*rho + e */
      tn = tn * rho_n + e_i;        /* unit response; temp_n applied below   abd.py:288 */
      dts = rj * dts + ts;
      ts = ts * rj + e_i + e_v;     /* tempinf + tempvac responses, unit boosts   abd.py:272, 378-386 */
      ci += inf[g];
      civ += inf[g] + vacs[o];
      un[o] = tn;
      dn[o] = dtn;
      ds[o] = dts * waner[j];
      cum_n[o] = ci > 0;            /* cumsum(exposure) > 0   abd.py:306 */
      cum_s[o] = civ > 0;
      mu_n[o] = (ci > 0 ? perm_n : 0.0) + temp_n * tn + init_n;  /* abd.py:341 */
      mu_s[o] = (civ > 0 ? perm_s : 0.0) + ts + init_s;          /* abd.py:389-391 */
      if (i_out) i_out[(size_t)g * N + j] = inf[g];
    }
  }

  double g[NT];
  double lp = priors(theta, G, (double)cells, (double)n1, (double)N, (double)m1, g);

  for (int ag = 0; ag < 2; ++ag) {
    const int64_t K = ag ? K_s : K_n;
    const int32_t* kg = ag ? s_gap : n_gap;
    const int32_t* ki = ag ? s_ind : n_ind;
    const double* kx = ag ? s_x : n_x;
    const double* ky = ag ? s_y : n_y;
    const double* mu = ag ? mu_s : mu_n;
    const int8_t* cum = ag ? cum_s : cum_n;
    const double b = ag ? b_s : b_n, d = ag ? d_s : d_n, sig = ag ? sig_s : sig_n;
    double ll = 0, gh = 0, ghc = 0, ghu = 0, ghd = 0, ghx = 0, gws = 0;
#pragma omp parallel for schedule(static) reduction(+ : ll, gh, ghc, ghu, ghd, ghx, gws)
    for (int64_t k = 0; k < K; ++k) {
      const size_t o = (size_t)ki[k] * G + kg[k]; /* mu[idx_gap, idx_ind]   abd.py:343, 393 */
      const double a = mu[o], x = kx[k], y = ky[k];
      const double e = exp(-b * (x - a)); /* logistic   abd.py:557 */
      const double s = 1.0 / (1.0 + e);
      const double r = (y - d * s) / sig;
      ll += r * r;
      const double w = r / sig;
      const double h = w * d * s * (e * s);
      gh += h;
      ghc += cum[o] ? h : 0.0;
      ghu += h * (ag ? 0.0 : un[o]);
      ghd += h * (ag ? ds[o] : dn[o]);
      ghx += h * (a - x);
      gws += w * s;
    }
    lp += -0.5 * ll - (double)K * (log(sig) + 0.5 * LOG_2PI);
    if (!ag) {
      g[1] += -b * perm_n * ghc;
      g[2] += -b * temp_n * ghu;
      g[3] += -b * temp_n * rho_n * (1 - rho_n) * ghd;
      g[4] += -b * gh;
      g[11] += -ghx;
      g[12] += gws;
      g[13] += ll - (double)K;
    } else {
      g[5] += -b * perm_s * ghc;
      g[6] += -b * rho_s * (1 - rho_s) * ghd;
      g[10] += -b * gh;
      g[14] += -ghx;
      g[15] += gws;
      g[16] += ll - (double)K;
    }
  }
  *logp = lp;
  if (grad) memcpy(grad, g, sizeof g);
  free(mu_n);
  free(mu_s);
  free(un);
  free(dn);
  free(ds);
  free(cums);
  return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Binary Gibbs-Metropolis sweep over [i_raw, ab_s_waner]: CPU restatement of abd_gibbs_kernel
 * (abdpymc_amd/csrc/abd_gibbs.hpp), same Philox4x32-10 stream, same order, same acceptance rule.
 * Semantics: PyMC BinaryGibbsMetropolis.astep (transit_p = 0.8, shuffled dims, metrop_select) on the two
 * discrete variables (abd.py:427, 373); the per-individual delta equals the joint-logp difference
 * (tests/test_gibbs.py checks that against full joint evaluations).
 * ------------------------------------------------------------------------------------------------- */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void abd_oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

typedef struct {
  int G;
  double perm_n, temp_n, rho_n, init_n, perm_s, rho_s, init_s, b_n, d_n, is2_n, b_s, d_s, is2_s;
} gibbs_par;

/* -1/2 sum (q/sigma)^2 of one individual's observations for infection column inf[], vaccinations vac[], waner w */
static double individual_ll(const gibbs_par* p, const int8_t* inf, const int8_t* vac, int w, int64_t ks0, int64_t ks1,
                            const int32_t* s_gap, const double* s_x, const double* s_y, int64_t kn0, int64_t kn1,
                            const int32_t* n_gap, const double* n_x, const double* n_y) {
  double mu_n[1024], mu_s[1024];
  const double rj = p->rho_s * w + 1 - w;
  double tn = 0, ts = 0;
  int ci = 0, civ = 0;
  for (int g = 0; g < p->G; ++g) {
    tn = tn * p->rho_n + inf[g];
    ts = ts * rj + inf[g] + vac[g];
    ci += inf[g];
    civ += inf[g] + vac[g];
    mu_n[g] = (ci > 0 ? p->perm_n : 0.0) + p->temp_n * tn + p->init_n;
    mu_s[g] = (civ > 0 ? p->perm_s : 0.0) + ts + p->init_s;
  }
  double acc = 0;
  for (int64_t k = kn0; k < kn1; ++k) {
    double q = n_y[k] - p->d_n / (1.0 + exp(-p->b_n * (n_x[k] - mu_n[n_gap[k]])));
    acc += -0.5 * p->is2_n * q * q;
  }
  for (int64_t k = ks0; k < ks1; ++k) {
    double q = s_y[k] - p->d_s / (1.0 + exp(-p->b_s * (s_x[k] - mu_s[s_gap[k]])));
    acc += -0.5 * p->is2_s * q * q;
  }
  return acc;
}

/* sort one antigen's observations by individual (stable): out arrays sized K, ptr sized N+1 */
static void csr_by_ind(int N, int64_t K, const int32_t* gap, const int32_t* ind, const double* x, const double* y,
                       int64_t* ptr, int32_t* ogap, double* ox, double* oy) {
  for (int j = 0; j <= N; ++j) ptr[j] = 0;
  for (int64_t k = 0; k < K; ++k) ptr[ind[k] + 1]++;
  for (int j = 0; j < N; ++j) ptr[j + 1] += ptr[j];
  int64_t* cur = (int64_t*)malloc((size_t)(N + 1) * sizeof(int64_t));
  memcpy(cur, ptr, (size_t)(N + 1) * sizeof(int64_t));
  for (int64_t k = 0; k < K; ++k) {
    int64_t o = cur[ind[k]]++;
    ogap[o] = gap[k]; ox[o] = x[k]; oy[o] = y[k];
  }
  free(cur);
}

int abd_oracle_gibbs_sweep(int G, int N, int n_splits, const int* splits, const int8_t* vacs, const int8_t* pcrpos,
                           int64_t K_s, const int32_t* s_gap, const int32_t* s_ind, const double* s_x, const double* s_y,
                           int64_t K_n, const int32_t* n_gap, const int32_t* n_ind, const double* n_x, const double* n_y,
                           int8_t* i_raw /* (G,N) in/out */, int8_t* waner /* (N) in/out */, const double* theta,
                           int chain, uint64_t seed, uint32_t sweep, int64_t* accepted, int64_t* proposed, int nthreads) {
  if (G < 2 || G > 1023 || N < 1 || n_splits < 0 || n_splits > 2) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  gibbs_par p;
  p.G = G;
  p.perm_n = exp(theta[1]); p.temp_n = exp(theta[2]); p.rho_n = sigmoid(theta[3]); p.init_n = theta[4];
  p.perm_s = exp(theta[5]); p.rho_s = sigmoid(theta[6]); p.init_s = theta[10];
  p.b_n = theta[11]; p.d_n = theta[12]; p.is2_n = 1.0 / (exp(theta[13]) * exp(theta[13]));
  p.b_s = theta[14]; p.d_s = theta[15]; p.is2_s = 1.0 / (exp(theta[16]) * exp(theta[16]));
  const uint32_t k0 = (uint32_t)seed ^ (sweep * 0x9E3779B9u), k1 = (uint32_t)(seed >> 32);
  int64_t* sp = (int64_t*)malloc((size_t)(N + 1) * 8); int64_t* np_ = (int64_t*)malloc((size_t)(N + 1) * 8);
  int32_t* sg = (int32_t*)malloc((size_t)(K_s + 1) * 4); int32_t* ng = (int32_t*)malloc((size_t)(K_n + 1) * 4);
  double* sx = (double*)malloc((size_t)(K_s + 1) * 8); double* sy = (double*)malloc((size_t)(K_s + 1) * 8);
  double* nx = (double*)malloc((size_t)(K_n + 1) * 8); double* ny = (double*)malloc((size_t)(K_n + 1) * 8);
  csr_by_ind(N, K_s, s_gap, s_ind, s_x, s_y, sp, sg, sx, sy);
  csr_by_ind(N, K_n, n_gap, n_ind, n_x, n_y, np_, ng, nx, ny);
  long acc_total = 0, prop_total = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : acc_total, prop_total)
  for (int j = 0; j < N; ++j) {
    int8_t raw[1024], rawn[1024], inf[1024], infn[1024], vac[1024];
    const int8_t* pcr = pcrpos ? pcrpos + (size_t)j * G : 0;
    for (int g = 0; g < G; ++g) { raw[g] = i_raw[(size_t)g * N + j]; vac[g] = vacs[(size_t)j * G + g]; }
    int w = waner[j];
    constrain_column(G, n_splits, splits, raw, pcr, inf);
    /* order: rank by (word0 & ~0x1FF) | dim */
    const int nd = G + 1;
    uint32_t key[1025]; int order[1025]; uint32_t tr[1025], ac[1025];
    for (int d = 0; d < nd; ++d) {
      uint32_t r[4];
      philox4x32_10((uint32_t)d, (uint32_t)j, (uint32_t)chain, 0u, k0, k1, r);
      key[d] = (r[0] & ~0x1FFu) | (uint32_t)d; tr[d] = r[1]; ac[d] = r[2];
    }
    for (int d = 0; d < nd; ++d) {
      int rank = 0;
      for (int e = 0; e < nd; ++e) rank += key[e] < key[d];
      order[rank] = d;
    }
    double ll = individual_ll(&p, inf, vac, w, sp[j], sp[j + 1], sg, sx, sy, np_[j], np_[j + 1], ng, nx, ny);
    for (int k = 0; k < nd; ++k) {
      const int d = order[k];
      if (!(tr[d] < 3435973836u)) continue; /* transit_p = 0.8 */
      prop_total++;
      double delta, ll_new = ll;
      int wn = w, same;
      memcpy(rawn, raw, (size_t)G);
      if (d < G) {
        delta = raw[d] ? -theta[0] : theta[0];
        rawn[d] ^= 1;
        constrain_column(G, n_splits, splits, rawn, pcr, infn);
        same = memcmp(inf, infn, (size_t)G) == 0;
      } else {
        wn = !w;
        delta = wn ? theta[7] : -theta[7];
        memcpy(infn, inf, (size_t)G);
        same = 0;
      }
      if (!same) {
        ll_new = individual_ll(&p, infn, vac, wn, sp[j], sp[j + 1], sg, sx, sy, np_[j], np_[j + 1], ng, nx, ny);
        delta += ll_new - ll;
      }
      const double u = ((double)ac[d] + 0.5) * (1.0 / 4294967296.0);
      if (delta > 0.0 || delta > log(u)) {
        memcpy(raw, rawn, (size_t)G); memcpy(inf, infn, (size_t)G);
        w = wn; ll = ll_new; acc_total++;
      }
    }
    for (int g = 0; g < G; ++g) i_raw[(size_t)g * N + j] = raw[g];
    waner[j] = (int8_t)w;
  }
  if (accepted) *accepted = acc_total;
  if (proposed) *proposed = prop_total;
  free(sp); free(np_); free(sg); free(ng); free(sx); free(sy); free(nx); free(ny);
  return 0;
}

int abd_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
