/* sia2d_oracle.c -- CPU restatement (C99 + OpenMP, fp64) of the SIA2D forward RHS, one
 * RDPK3Sp35 step and the discrete H-VJP of ODINN.jl.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/ and the `cpu_baseline` leg of bench.py.
 * Nothing in the product path (odinn.jl_amd/) links or calls it.
 *
 * PARITY STATUS: forward-value parity unpinned (no Julia toolchain in the build image; the
 * forward kernel Huginn.SIA2D! is an un-vendored dependency).  The arithmetic follows the
 * reference's own discrete adjoint, which re-executes the forward stencil:
 *   forward intermediates     src/inverse/SIA2D/adjoint.jl:52-97
 *   border clamp              src/inverse/SIA2D/inversion_utils.jl:17-20,31-34
 *   diffusivity (:A target)   src/models/target/target_A.jl:16-62, target_utils.jl:3-18
 *   adjoint                   src/inverse/SIA2D/adjoint.jl:99-148, inversion_utils.jl:3-66
 * and is cross-checked against the numpy oracle (oracle/sia2d_oracle.py) in
 * tests/test_oracle_c.py.
 *
 * Layout: element [i,j] at i + nx*j (Julia column-major, i = x contiguous).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double rho, g, eta0, n, p, q, C;
} oc_phys;

static inline double clampd(double e, double up, double lo) { return fmax(fmin(e, up), lo); }

int oc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void oc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* D on the dual grid (nx-1)*(ny-1); Hc = max(H,0), S = B+Hc are filled too (adjoint.jl:52-84) */
/* Af != NULL: A on the dual grid (a gridded LawA hoisted once per theta, Laws.jl:339-358: A.value broadcasts against Hbar in
 * target_A.jl:25-29), read per node instead of the scalar A */
static void oc_diffusivity(int nx, int ny, const double* H, const double* B, double dx, double dy, const oc_phys* ph,
                           double A0, const double* Af, double* Hc, double* S, double* D) {
  const double Gam = 2.0 * pow(ph->rho * ph->g, ph->n) / (ph->n + 2.0);
  const double Sc = ph->C * pow(ph->rho * ph->g, ph->p - ph->q);
  const int fast = (ph->n == 3.0 && Sc == 0.0);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      const double h = H[i + (size_t)nx * j];
      const double hc = h > 0.0 ? h : 0.0;
      Hc[i + (size_t)nx * j] = hc;
      S[i + (size_t)nx * j] = B[i + (size_t)nx * j] + hc;
    }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ny - 1; ++j)
    for (int i = 0; i < nx - 1; ++i) {
      const size_t a = i + (size_t)nx * j;
      const double gx = 0.5 * ((S[a + 1] - S[a]) / dx + (S[a + nx + 1] - S[a + nx]) / dx);
      const double gy = 0.5 * ((S[a + nx] - S[a]) / dy + (S[a + nx + 1] - S[a + 1]) / dy);
      const double Hb = 0.25 * (Hc[a] + Hc[a + 1] + Hc[a + nx] + Hc[a + nx + 1]);
      const double g2 = gx * gx + gy * gy;
      const double A = Af ? Af[i + (size_t)(nx - 1) * j] : A0;
      double d;
      if (fast) {
        const double h2 = Hb * Hb;
        d = A * Gam * (h2 * h2 * Hb) * g2;
      } else {
        const double gs = sqrt(g2);
        d = A * Gam * pow(Hb, ph->n + 2.0) * pow(gs, ph->n - 1.0);
        if (Sc != 0.0) d += Sc * pow(Hb, ph->p - ph->q + 1.0) * pow(gs, ph->p - 1.0);
      }
      D[i + (size_t)(nx - 1) * j] = d;
    }
}

/* dH = SIA2D(H)  (Huginn.SIA2D!, restated from adjoint.jl:52-97).  work: 3*nx*ny doubles */
static void oc_rhs_impl(int nx, int ny, const double* H, const double* B, double dx, double dy, const oc_phys* ph, double A,
                        const double* Af, double* dH, double* work) {
  double* Hc = work;
  double* S = work + (size_t)nx * ny;
  double* D = work + 2 * (size_t)nx * ny;
  oc_diffusivity(nx, ny, H, B, dx, dy, ph, A, Af, Hc, S, D);
  const double e0 = ph->eta0;
  const int nd = nx - 1;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      const size_t c = i + (size_t)nx * j;
      if (i == 0 || j == 0 || i == nx - 1 || j == ny - 1) { dH[c] = 0.0; continue; }
      const double S0 = S[c], H0 = Hc[c];
      const double Dsw = D[(i - 1) + (size_t)nd * (j - 1)], Dse = D[i + (size_t)nd * (j - 1)];
      const double Dnw = D[(i - 1) + (size_t)nd * j], Dne = D[i + (size_t)nd * j];
      const double ee = clampd((S[c + 1] - S0) / dx, e0 * Hc[c + 1] / dx, -e0 * H0 / dx);
      const double ew = clampd((S0 - S[c - 1]) / dx, e0 * H0 / dx, -e0 * Hc[c - 1] / dx);
      const double en = clampd((S[c + nx] - S0) / dy, e0 * Hc[c + nx] / dy, -e0 * H0 / dy);
      const double es = clampd((S0 - S[c - nx]) / dy, e0 * H0 / dy, -e0 * Hc[c - nx] / dy);
      const double Fe = -(0.5 * (Dse + Dne)) * ee, Fw = -(0.5 * (Dsw + Dnw)) * ew;
      const double Fn = -(0.5 * (Dnw + Dne)) * en, Fs = -(0.5 * (Dsw + Dse)) * es;
      dH[c] = -((Fe - Fw) / dx + (Fn - Fs) / dy);
    }
}

void oc_sia2d_rhs(int nx, int ny, const double* H, const double* B, double dx, double dy, const oc_phys* ph, double A,
                  double* dH, double* work) {
  oc_rhs_impl(nx, ny, H, B, dx, dy, ph, A, NULL, dH, work);
}
/* the same with A given on the dual grid ((nx-1)*(ny-1), element [i,j] at i + (nx-1)*j) */
void oc_sia2d_rhs_field(int nx, int ny, const double* H, const double* B, double dx, double dy, const oc_phys* ph,
                        const double* Afield, double* dH, double* work) {
  oc_rhs_impl(nx, ny, H, B, dx, dy, ph, 0.0, Afield, dH, work);
}

/* RDPK3Sp35 (Ranocha et al. 2022), 3S*+ registers */
static const double G1[5] = {0.0, 2.587771979725733308135192812685323706e-01, -1.324380360140723382965420909764953437e-01,
                             5.056033948190826045833606441415585735e-02, 5.670532000739313812633197158607642990e-01};
static const double G2[5] = {1.0, 5.528354909301389892439698870483746541e-01, 6.731871608203061824849561782794643600e-01,
                             2.803103963297672407841316576323901761e-01, 5.521525447020610386070346724931300367e-01};
static const double G3[5] = {0.0, 0.0, 0.0, 2.752563273304676380891217287572780582e-01,
                             -8.950526174674033822276061734289327568e-01};
static const double DL[5] = {1.0, 3.407655879334525365094815965895763636e-01, 3.414382655003386206551709871126405331e-01,
                             7.229275366787987419692007421895451953e-01, 0.0};
static const double BT[5] = {2.300298624518076223899418286314123354e-01, 3.021434166948288809034402119555380003e-01,
                             8.025606185416310937583009085873554681e-01, 4.362158943603440930655148245148766471e-01,
                             1.129272530455059129782111662594436580e-01};
static const double BH[5] = {1.046363371354093758897668305991705199e-01, 9.520431574956758809511173383346476348e-02,
                             4.482446645568668405072421350300379357e-01, 2.449030295461310135957132640369862245e-01,
                             1.070116530120251819121660365003405564e-01};

/* one step u -> u (in place); returns the scaled RMS error estimate.
 * work: 7*nx*ny doubles (k, tmp, uprev, utilde + 3 for the RHS) */
static double oc_step_impl(int nx, int ny, double* u, const double* B, double dx, double dy, const oc_phys* ph, double A,
                           const double* Af, double dt, double abstol, double reltol, double* work) {
  const size_t N = (size_t)nx * ny;
  double* k = work;
  double* tmp = work + N;
  double* up = work + 2 * N;
  double* ut = work + 3 * N;
  double* w2 = work + 4 * N;
  memcpy(up, u, N * sizeof(double));
  memcpy(tmp, u, N * sizeof(double));
  oc_rhs_impl(nx, ny, u, B, dx, dy, ph, A, Af, k, w2);
#pragma omp parallel for schedule(static)
  for (size_t c = 0; c < N; ++c) {
    u[c] = tmp[c] + BT[0] * dt * k[c];
    ut[c] = BH[0] * dt * k[c];
  }
  for (int s = 1; s < 5; ++s) {
    oc_rhs_impl(nx, ny, u, B, dx, dy, ph, A, Af, k, w2);
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < N; ++c) {
      tmp[c] = tmp[c] + DL[s] * u[c];
      u[c] = G1[s] * u[c] + G2[s] * tmp[c] + G3[s] * up[c] + BT[s] * dt * k[c];
      ut[c] = ut[c] + BH[s] * dt * k[c];
    }
  }
  double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (size_t c = 0; c < N; ++c) {
    const double err = (u[c] - up[c]) - ut[c];
    const double sk = abstol + fmax(fabs(up[c]), fabs(u[c])) * reltol;
    acc += (err / sk) * (err / sk);
  }
  return sqrt(acc / (double)N);
}
double oc_rdpk3sp35_step(int nx, int ny, double* u, const double* B, double dx, double dy, const oc_phys* ph, double A,
                         double dt, double abstol, double reltol, double* work) {
  return oc_step_impl(nx, ny, u, B, dx, dy, ph, A, NULL, dt, abstol, reltol, work);
}
double oc_rdpk3sp35_step_field(int nx, int ny, double* u, const double* B, double dx, double dy, const oc_phys* ph,
                               const double* Afield, double dt, double abstol, double reltol, double* work) {
  return oc_step_impl(nx, ny, u, B, dx, dy, ph, 0.0, Afield, dt, abstol, reltol, work);
}

/* discrete H-VJP, written as the reference writes it: explicit scatter-style transposes on
 * full scratch arrays (adjoint.jl:99-148, inversion_utils.jl:3-66).  work: 14*nx*ny doubles */
void oc_vjp_H(int nx, int ny, const double* lam, const double* H, const double* B, double dx, double dy,
              const oc_phys* ph, double A, double* dlam, double* work) {
  const size_t N = (size_t)nx * ny;
  const int ndx = nx - 1, ndy = ny - 1;
  double* Hc = work;            double* S = work + N;        double* D = work + 2 * N;
  double* gSx = work + 3 * N;   double* gSy = work + 4 * N;  double* al = work + 5 * N;
  double* be = work + 6 * N;    double* Da = work + 7 * N;   double* Fxa = work + 8 * N;
  double* Fya = work + 9 * N;   double* exs = work + 10 * N; double* eys = work + 11 * N;
  double* T = work + 12 * N;    double* U = work + 13 * N;
  oc_diffusivity(nx, ny, H, B, dx, dy, ph, A, NULL, Hc, S, D);
  const double Gam = 2.0 * pow(ph->rho * ph->g, ph->n) / (ph->n + 2.0);
  const double Sc = ph->C * pow(ph->rho * ph->g, ph->p - ph->q);
  const double e0 = ph->eta0;
  for (int j = 0; j < ndy; ++j)
    for (int i = 0; i < ndx; ++i) {
      const size_t a = i + (size_t)nx * j, d = i + (size_t)ndx * j;
      const double gx = 0.5 * ((S[a + 1] - S[a]) / dx + (S[a + nx + 1] - S[a + nx]) / dx);
      const double gy = 0.5 * ((S[a + nx] - S[a]) / dy + (S[a + nx + 1] - S[a + 1]) / dy);
      const double Hb = 0.25 * (Hc[a] + Hc[a + 1] + Hc[a + nx] + Hc[a + nx + 1]);
      const double gs = sqrt(gx * gx + gy * gy);
      gSx[d] = gx; gSy[d] = gy;
      al[d] = A * Gam * (ph->n + 2.0) * pow(Hb, ph->n + 1.0) * pow(gs, ph->n - 1.0);
      be[d] = A * Gam * (ph->n - 1.0) * pow(Hb, ph->n + 2.0) * pow(gs, ph->n - 3.0);
      if (Sc != 0.0) {
        al[d] += (ph->p - ph->q + 1.0) * Sc * pow(Hb, ph->p - ph->q) * pow(gs, ph->p - 1.0);
        be[d] += Sc * (ph->p - 1.0) * pow(Hb, ph->p - ph->q + 1.0) * pow(gs, ph->p - 3.0);
      }
      Da[d] = 0.0;
    }
  /* x edges (i in 0..nx-2, j in 1..ny-2): Fxa = diff_x_adjoint(-lam_inn), clamp, D_adjoint */
  for (int j = 1; j <= ny - 2; ++j)
    for (int i = 0; i <= nx - 2; ++i) {
      const size_t c = i + (size_t)nx * j;
      const double li = (i >= 1) ? lam[c] : 0.0, lip = (i + 1 <= nx - 2) ? lam[c + 1] : 0.0;
      const double fa = (lip - li) / dx;
      const double e = (S[c + 1] - S[c]) / dx;
      const double ec = clampd(e, e0 * Hc[c + 1] / dx, -e0 * Hc[c] / dx);
      Fxa[c] = fa; exs[c] = e;
      Da[i + (size_t)ndx * (j - 1)] += 0.5 * (-fa * ec);
      Da[i + (size_t)ndx * j] += 0.5 * (-fa * ec);
    }
  for (int j = 0; j <= ny - 2; ++j)
    for (int i = 1; i <= nx - 2; ++i) {
      const size_t c = i + (size_t)nx * j;
      const double lj = (j >= 1) ? lam[c] : 0.0, ljp = (j + 1 <= ny - 2) ? lam[c + nx] : 0.0;
      const double fa = (ljp - lj) / dy;
      const double e = (S[c + nx] - S[c]) / dy;
      const double ec = clampd(e, e0 * Hc[c + nx] / dy, -e0 * Hc[c] / dy);
      Fya[c] = fa; eys[c] = e;
      Da[(i - 1) + (size_t)ndx * j] += 0.5 * (-fa * ec);
      Da[i + (size_t)ndx * j] += 0.5 * (-fa * ec);
    }
  memset(T, 0, N * sizeof(double));
  /* first term: avg_adjoint(alpha*Da) + diff_x_adjoint(avg_y_adjoint(bx*Da)) + diff_y_adjoint(avg_x_adjoint(by*Da)) */
  for (int j = 0; j < ndy; ++j)
    for (int i = 0; i < ndx; ++i) {
      const size_t a = i + (size_t)nx * j, d = i + (size_t)ndx * j;
      const double ad = 0.25 * al[d] * Da[d];
      const double bx = 0.5 * be[d] * gSx[d] * Da[d] / dx, by = 0.5 * be[d] * gSy[d] * Da[d] / dy;
      T[a] += ad - bx - by;
      T[a + 1] += ad + bx - by;
      T[a + nx] += ad - bx + by;
      T[a + nx + 1] += ad + bx + by;
    }
  /* second term: clamp adjoint (inversion_utils.jl:22-29,36-43) */
  memset(U, 0, N * sizeof(double));
  for (int j = 1; j <= ny - 2; ++j)
    for (int i = 0; i <= nx - 2; ++i) {
      const size_t c = i + (size_t)nx * j;
      const double Dx = 0.5 * (D[i + (size_t)ndx * (j - 1)] + D[i + (size_t)ndx * j]);
      const double C = -Fxa[c] * Dx;
      const double up = e0 * Hc[c + 1] / dx, lo = -e0 * Hc[c] / dx, e = exs[c];
      if (e < up && e > lo) { U[c + 1] += C / dx; U[c] -= C / dx; }
      if (e < lo) U[c] += -(e0 * C / dx);
      if (e > up) U[c + 1] += (e0 * C / dx);
    }
  for (int j = 0; j <= ny - 2; ++j)
    for (int i = 1; i <= nx - 2; ++i) {
      const size_t c = i + (size_t)nx * j;
      const double Dy = 0.5 * (D[(i - 1) + (size_t)ndx * j] + D[i + (size_t)ndx * j]);
      const double C = -Fya[c] * Dy;
      const double up = e0 * Hc[c + nx] / dy, lo = -e0 * Hc[c] / dy, e = eys[c];
      if (e < up && e > lo) { U[c + nx] += C / dy; U[c] -= C / dy; }
      if (e < lo) U[c] += -(e0 * C / dy);
      if (e > up) U[c + nx] += (e0 * C / dy);
    }
  for (size_t c = 0; c < N; ++c) dlam[c] = (Hc[c] > 0.0) ? (T[c] + U[c]) : 0.0;
}

/* G independent glaciers advanced `nsteps` RDPK3Sp35 steps each, ONE THREAD PER GLACIER (the inner
 * `parallel for`s run serially inside this parallel region: nested parallelism is off).  This is how
 * the reference uses a multi-core host -- one glacier per worker process (pmap, gradient.jl:9-10,
 * src/setup/config.jl:97-139) -- and it is the CPU baseline bench.py reports.
 * u, work: arrays of G pointers (nx*ny and 7*nx*ny doubles each); B shared. */
void oc_multi_steps(int G, int nsteps, int nx, int ny, double** u, const double* B, double dx, double dy,
                    const oc_phys* ph, double A, double dt, double** work) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int g = 0; g < G; ++g)
    for (int s = 0; s < nsteps; ++s) oc_rdpk3sp35_step(nx, ny, u[g], B, dx, dy, ph, A, dt, 1e-6, 1e-8, work[g]);
}
/* the same with a dual-grid A field shared by the G copies (bench.py's cpu_baseline on the headline's workload: the GPU reads
 * the hoisted A = NN_theta(T) field in every stage, +8 B per cell-stage, and so does this) */
void oc_multi_steps_field(int G, int nsteps, int nx, int ny, double** u, const double* B, double dx, double dy,
                          const oc_phys* ph, const double* Afield, double dt, double** work) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int g = 0; g < G; ++g)
    for (int s = 0; s < nsteps; ++s) oc_step_impl(nx, ny, u[g], B, dx, dy, ph, 0.0, Afield, dt, 1e-6, 1e-8, work[g]);
}
