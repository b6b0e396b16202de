// sia2d_adj_fused.hpp -- one whole RDPK3Sp35 step of the REVERSE ODE of the continuous adjoint in ONE kernel
//     dlam/dtau = J_H(H_itp(-tau))^T lam                     (gradient.jl:316-324, adjoint.jl:99-148)
// for the integer-power A law with the DiscreteVJP stencil: the temporal fusion of k_adj_stage<1..5>, built
// like k_rk_fused_strip (sia2d_fused.hpp): a wavefront owns TRPT contiguous region rows, a thread one column
// of them; y-neighbours are the thread's own registers, x-neighbours arrive by DPP wave shifts, only the
// first and last row of a strip cross wavefronts through a double-buffered LDS exchange (one barrier per
// stage).  HBM traffic per step: read lam, H_j, H_j+1, B, write lam' = 40 B/cell instead of 5 x 72.
//
// Face form of the H-VJP.  k_vjp_H evaluates adjoint.jl:99-148 node by node (vjpH_node: every dual node
// handles its four edges, so each edge is visited twice).  Here a thread owns, per row, its cell, the EAST and
// NORTH face of the cell and the node at the cell's north-east corner:
//   face e(c,r):  dS = S(c+1,r) - S(c,r),  q = lam(c+1,r) - lam(c,r)   (lam masked to the interior, the
//                 transposed difference of adjoint.jl:100-101),  P = q clamp(dS; H(c+1,r), -H(c,r))
//   node N(c,r):  Da = -(P_e(c,r) + P_e(c,r+1)) / (2 dx^2) - (P_n(c,r) + P_n(c+1,r)) / (2 dy^2)    (:102-104)
//                 corner terms  alpha Da / 4 -+ beta gx Da / (2 dx) -+ beta gy Da / (2 dy)         (:123-127)
//   face e(c,r), second term (:130-144, inversion_utils.jl:22-43):  t = (D(c,r-1) + D(c,r)) q / (2 dx^2);
//                 +t to the minus cell (c,r) unless dS >= H+ or dS == -H-;  -t to the plus cell (c+1,r) unless
//                 dS <= -H- or dS == H+   (strict inequalities of the reference; eta0 == 1 in this law mode)
// and a cell gathers: its own node's SW term, the west lane's SE term and east-face plus part (one DPP of their
// sum), the NW / NE terms and the north-face plus part of the row below (carried up the sweep).  The result is
// masked by H > 0 (:147-148).  Per cell the same quantities as vjpH_node with the two visits of an edge
// merged; rounding differs in where products are summed, nothing else.
#pragma once
#include "sia2d_fused.hpp"

namespace odinn {

// {Hc,S} and masked lambda of the first | last row of each wavefront's strip, double-buffered per stage
typedef double2 (*AdjEdgesHS)[TNW][2][FRX];
typedef double (*AdjEdgesL)[TNW][2][FRX];

// ODINN_ADJ_ELDS: the embedded-error accumulator of the thread's rows lives in a thread-private LDS column instead of 2 NR
// VGPRs (it is touched once per row and stage)
#define ODINN_ADJ_ELDS 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
// ODINN_ADJ_PF2: constant-A variants fetch {Hc,S} two rows ahead instead of one (fits 128 VGPRs without scratch once E is in
// LDS).  Same-box A/B: +2.1 % on the bench's continuous-adjoint gradient (half ice-free domains, shortcut on), but -10 % on
// the dense all-ice launch (0.260 -> 0.287 ms per 8 x 1024^2): off.
#define ODINN_ADJ_PF2 0  // (fixed: its A/B is recorded above; no longer a build-time knob)
// ODINN_ADJ_APF: gridded A one node row ahead of its use (like {Hc,S}) instead of inside node_face (8 x 1024^2, gridded
// law: 282 -> 262 us per reverse step; 0 restores the load at the point of use)
#define ODINN_ADJ_APF 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
// ODINN_ADJ_RC: the gridded-A variants keep everything a stage re-reads -- {H_j, H_j+1 - H_j}, B, the A nodes and lambda at the
// start of the step -- in registers for the five stages (4 doubles per row + the A row below the strip: 72 VGPRs at 7 rows) and
// run at 2 wavefronts per SIMD (one workgroup per CU, 256 VGPRs).  At 4 per SIMD the variant spilled 11-17 registers and was
// HBM-bound on its own re-reads: PMC at 64 x 1024^2, 15.4 GB fetched per launch = 5 x the algorithmic traffic at 5.4 TB/s
// (3.05 ms; the constant-A variant: 6.3 GB, 2.13 ms) -- the A field is one more 512 MB stream through a 4 MB L2 in every stage.
#define ODINN_ADJ_RC 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
#define ODINN_ADJ_RC_EREG 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
typedef double (*AdjErr)[FRX];
// register cache of a thread's stage-invariant inputs (ODINN_ADJ_RC)
template <int NR>
struct AdjRowCache {
  double2 hd[NR];     // {H_j, H_j+1 - H_j}
  double b[NR];       // bed
  double An[NR + 1];  // A at the node rows gj0 + r0 - 1 ... gj0 + r0 + NR - 1
  double u0[NR];      // lambda at the start of the step
};

template <int S, bool AF, bool SG, int NR, bool GA = false, bool RC = false, bool YT = false, bool UT = false>
__device__ __forceinline__ void adj_strip_stage(const GDev& g, const double* __restrict__ Afield, const double* __restrict__ Ha,
                                                 const double* __restrict__ Hb, const double* __restrict__ src, const AdjState& a,
                                                 int gic, int gi, int gj0, int w, int lane, double dt, AdjEdgesHS sE,
                                                 AdjEdgesL sLm, double (&u)[NR], double (&tmp)[NR], double (&E)[NR],
                                                 const double* __restrict__ Bp, AdjErr sEr, double* __restrict__ th_red,
                                                 [[maybe_unused]] double* __restrict__ Gp = nullptr,
                                                 [[maybe_unused]] const AdjRowCache<RC ? NR : 1>* rc = nullptr,
                                                 [[maybe_unused]] const YtabRef yt = YtabRef{nullptr, nullptr, 0, nullptr},
                                                 [[maybe_unused]] double* __restrict__ Eh = nullptr,
                                                 [[maybe_unused]] double* __restrict__ Ev = nullptr,
                                                 [[maybe_unused]] double* __restrict__ emax = nullptr,
                                                 [[maybe_unused]] const LawDev* Lu = nullptr) {
  constexpr int rd = (S - 1) & 1, wr = S & 1;
  constexpr bool ELDS = ODINN_ADJ_ELDS && !(RC && ODINN_ADJ_RC_EREG);
  const int r0 = NR * w;
  [[maybe_unused]] const bool nodex = gi >= 0 && gi <= g.nx - 2;
  const bool intx = gi >= 1 && gi <= g.nx - 2;
  constexpr int s = S - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  const double Gq = g.Gam * (1.0 / 1024.0);  // A Gam Hbar^k = (A Gam / 1024)(4 Hbar)^k scaled by exact powers of two
  const double sw = a.sitp[S - 1];
  // Ha, Hb, src, Bp point at the glacier's first cell (block-uniform: scalar base registers); a cell is addressed by a
  // 32-bit index.  gif: the clamped column again, but opaque after every row fence -- keeps each row's loads in its row
  int gif = gic;
  // {Hc, S} of one of the thread's rows at a stage time: H_itp = H_j + sw (H_j+1 - H_j) (load_tile_HS2's formula)
  // and B re-read from global memory (L2-resident).  Outside the grid the index is clamped and whatever it picks
  // up is never used: lambda is masked to the interior, so every term that touches such a cell carries a factor 0.
  auto hs_itp = [&](int m, double swt) {
    if constexpr (RC) return cell_HS(fma(swt, rc->hd[m].y, rc->hd[m].x), rc->b[m]);
    const int gj = gj0 + r0 + m;
    const int gjc = gj < 0 ? 0 : (gj > g.ny - 1 ? g.ny - 1 : gj);
    const unsigned id = (unsigned)(gif + g.nx * gjc);  // zero-extended: scalar base + 32-bit offset addressing
    if constexpr (SG) {  // Ha points at the segment's {H_j, H_j+1 - H_j} pairs
      const double2 hd = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Ha) + ((size_t)id << 4));
      return cell_HS(fma(swt, hd.y, hd.x), ldg32(Bp, id));
    } else {
      const double ha = ldg32(Ha, id), hb = ldg32(Hb, id), b = ldg32(Bp, id);
      return cell_HS(fma(swt, hb - ha, ha), b);
    }
  };
  auto lam_e = [&](int m) {  // lambda masked to the interior cells
    const int gj = gj0 + r0 + m;
    return (intx && gj >= 1 && gj <= g.ny - 2) ? u[m] : 0.0;
  };
  // the rows just outside the strip (the outermost wavefronts read their own edge: rows 0 and (NR * TNW)-1 are never in region_S)
  const int wb = w > 0 ? w - 1 : 0, eb = w > 0 ? 1 : 0, wt = w + 1 < TNW ? w + 1 : w, et = w + 1 < TNW ? 0 : 1;
  const double2 hs_s = sE[rd][wb][eb][lane], hs_top = sE[rd][wt][et][lane];
  const double le_s = sLm[rd][wb][eb][lane], le_top = sLm[rd][wt][et][lane];
  // carried up the sweep (suffix _c: the row being processed)
  double2 hs_c = hs_itp(0, sw);
  double le_c = lam_e(0);
  double2 e_c = dpp_from_east(hs_c);
  double dx_c = e_c.y - hs_c.y, hp_c = hs_c.x + e_c.x, qe_c = dpp_shift(le_c, false) - le_c;
  double Pe_c = qe_c * clampn(dx_c, e_c.x, hs_c.x);
  double D_s, C_s;  // C_s: what the row below holds for this cell (its nodes' NW / NE terms, its north face's plus part)
  double2 hs_next = hs_itp(NR > 1 ? 1 : 0, sw);  // {Hc,S} of row m+1, fetched one row ahead of its use
  [[maybe_unused]] double thacc = 0.0;  // stage 1, th_red != null: the wavefront's share of the theta-VJP (see node_face)
  // constant A: registers allow a second row in flight ({Hc,S} of row m+2); with a gridded A they do not (96 B/lane of scratch)
  constexpr bool PF2 = ODINN_ADJ_PF2 && !AF;
  [[maybe_unused]] double2 hs_next2;
  if constexpr (PF2) hs_next2 = hs_itp(NR > 2 ? 2 : 0, sw);

  // node N(c, r) and north face n(c, r) of a row whose own / east-face quantities are the "_lo" arguments and whose
  // upper neighbours are the "_hi" ones; returns D, the four corner terms and the north face's two second-term parts
  auto node_face = [&](int gj, double2 hs_lo, double2 e_lo, double le_lo, double dx_lo, double hp_lo, double Pe_lo,
                       double2 hs_hi, double2 e_hi, double le_hi, double dx_hi, double hp_hi, double Pe_hi, double& D,
                       double& k00, double& k10, double& k01, double& k11, double& Mn, double& PLn, double& tw,
                       [[maybe_unused]] double An_pf, [[maybe_unused]] double* hbar = nullptr) {
    const double dyw = hs_hi.y - hs_lo.y, dye = e_hi.y - e_lo.y;
    const double qn = le_hi - le_lo;
    const double Pn = qn * clampn(dyw, hs_hi.x, hs_lo.x);
    const double Pn_e = dpp_shift(Pn, false);
    const double gx = (dx_lo + dx_hi) * g.hinv_dx, gy = (dyw + dye) * g.hinv_dy;
    const double Hs = hp_lo + hp_hi;  // 4 Hbar
    if constexpr (YT) { if (S == 1 && hbar) *hbar = 0.25 * Hs; }
    const double gS2 = gx * gx + gy * gy;
    double An = g.A;
    if (AF) {
      if constexpr (ODINN_ADJ_APF) {
        An = An_pf;
      } else {
        const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
        An = ldg32(Afield, (unsigned)(ok ? gif + (g.nx - 1) * gj : 0));  // Afield: the glacier's first dual node (ok: gif == gi)
      }
    }
    // YT: the Y law of target :D_hybrid with n_H = n_gradS = 3 and no sliding is this law with Y(Hbar) in A's place (node_D<LM_YTAB>,
    // yt_fast), plus the reference's finite-difference term of dD/dHbar (target_D_hybrid.jl:58-71) in alpha below
    [[maybe_unused]] double Yp = 0.0;
    if constexpr (YT) An = ytab_eval_acc<true>(yt.tab, yt.ni, *yt.beyond, g.yt_inv_h, 0.25 * Hs, Yp);  // (no branch inside the stage code)
    const double Da = -fma(g.hinv_dx2, Pe_lo + Pe_hi, g.hinv_dy2 * (Pn + Pn_e));
    if constexpr (UT) {
      // target :D (target_D_pure.jl:78-137): D = Hbar U, alpha = dD/dHbar and beta = dD/d|grad S| by central differences of
      // D with steps 1e-4 / 1e-6, the perturbed values of U from the node's own bi-quintic patch (node_D<LM_UTAB>'s arithmetic);
      // beta is NOT divided by |grad S| (as written upstream: "for now we ignore the derivative in surface slope")
      const double Hb = 0.25 * Hs;
      double al = 0.0, be = 0.0;
      D = 0.0;
      if (Hb > 0.0) {
        double up[4];
        const double U = utab_eval<true>(*Lu, Hb, sqrt(gS2), up);
        const double dH = 1e-4, dS = 1e-6;
        al = (up[0] * (Hb + dH) - up[1] * (Hb - dH)) / (2.0 * dH);
        be = (up[2] * Hb - up[3] * Hb) / (2.0 * dS);
        D = Hb * U;
      }
      tw = 0.0;
      const double ad = (0.25 * al) * Da, bd = be * Da;
      const double bx = g.hinv_dx * (bd * gx), by = g.hinv_dy * (bd * gy);
      const double am = ad - bx, ap = ad + bx;
      k00 = am - by; k10 = ap - by; k01 = am + by; k11 = ap + by;
      const double Dw = dpp_from_west(D);
      const double tn = ((Dw + D) * g.hinv_dy2) * qn;
      Mn = (dyw < hs_hi.x && dyw != -hs_lo.x) ? tn : 0.0;
      PLn = (dyw > -hs_lo.x && dyw != hs_hi.x) ? -tn : 0.0;
      return;
    }
    const double Kq = An * Gq;
    const double H2 = Hs * Hs, H4 = H2 * H2, H5 = H4 * Hs;
    D = (Kq * H5) * gS2;
    // (stage 1 only) the node's weight in the theta-VJP of the A-type laws, dD/dA x D_adjoint = Gam Hbar^5 |grad S|^2 Da
    // (adjoint.jl:235-250; k_vjp_theta_strip's expression): stage 1 sits exactly on the state the reverse solve has just
    // reached, so a quadrature node reached by the previous step gets its theta-VJP here instead of in a launch of its own
    tw = S == 1 ? ((Gq * H5) * gS2) * Da : 0.0;
    double ad = (((Kq * 5.0) * H4) * gS2) * Da;  // alpha Da / 4
    if constexpr (YT) {
      const double geo = (Gq * H5) * gS2;  // Gam Hbar^5 |grad S|^2
      // (the reference's quotient (D(Hbar + 1e-4) - D(Hbar)) / 1e-4 as a product: 1 / 1e-4 rounded once instead of a division per node
      //  and stage -- ~30 instructions; the quotient changes in its last bit)
      ad = fma(0.25 * ((Yp * geo - An * geo) * (1.0 / 1e-4)), Da, ad);
    }
    const double bd = ((Kq * 2.0) * H5) * Da;           // beta Da
    const double bx = g.hinv_dx * (bd * gx), by = g.hinv_dy * (bd * gy);
    const double am = ad - bx, ap = ad + bx;
    k00 = am - by; k10 = ap - by; k01 = am + by; k11 = ap + by;
    // north face, second term
    const double Dw = dpp_from_west(D);
    const double tn = ((Dw + D) * g.hinv_dy2) * qn;
    Mn = (dyw < hs_hi.x && dyw != -hs_lo.x) ? tn : 0.0;
    PLn = (dyw > -hs_lo.x && dyw != hs_hi.x) ? -tn : 0.0;
  };

  [[maybe_unused]] auto ld_A = [&](int k) {  // A at the node row gj0 + r0 + k, k = -1 ... NR - 1
    if constexpr (RC) return rc->An[k + 1];
    const int gj = gj0 + r0 + k;
    const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
    return ldg32(Afield, (unsigned)(ok ? gif + (g.nx - 1) * gj : 0));
  };
  [[maybe_unused]] double A_c = 0.0, A_s = 0.0;
  if constexpr (AF && ODINN_ADJ_APF) { A_s = ld_A(-1); A_c = ld_A(0); }
  {  // the node row and the north faces just below the strip
    const double2 e_s = dpp_from_east(hs_s);
    const double lee_s = dpp_shift(le_s, false);
    const double dx_s = e_s.y - hs_s.y, hp_s = hs_s.x + e_s.x;
    const double Pe_s = (lee_s - le_s) * clampn(dx_s, e_s.x, hs_s.x);
    double k00, k10, k01, k11, Mn, PLn, tw;
    node_face(gj0 + r0 - 1, hs_s, e_s, le_s, dx_s, hp_s, Pe_s, hs_c, e_c, le_c, dx_c, hp_c, Pe_c, D_s, k00, k10, k01, k11, Mn, PLn, tw, A_s);
    C_s = (k01 + dpp_from_west(k11)) + PLn;
  }
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int gj = gj0 + r0 + m;
    const double2 hs_n = m + 1 < NR ? hs_next : hs_top;
    if constexpr (PF2) {
      hs_next = hs_next2;
      if (m + 3 < NR) hs_next2 = hs_itp(m + 3 < NR ? m + 3 : m, sw);
    } else {
      if (m + 2 < NR) hs_next = hs_itp(m + 2 < NR ? m + 2 : m, sw);
    }
    [[maybe_unused]] double A_n = 0.0;
    if constexpr (AF && ODINN_ADJ_APF) { if (m + 1 < NR) A_n = ld_A(m + 1 < NR ? m + 1 : m); }
    const double le_n = m + 1 < NR ? lam_e(m + 1 < NR ? m + 1 : m) : le_top;
    const double2 e_n = dpp_from_east(hs_n);
    const double lee_n = dpp_shift(le_n, false);
    const double dx_n = e_n.y - hs_n.y, hp_n = hs_n.x + e_n.x, qe_n = lee_n - le_n;
    const double Pe_n = qe_n * clampn(dx_n, e_n.x, hs_n.x);
    double D_c, k00, k10, k01, k11, Mn, PLn, tw;
    [[maybe_unused]] double hbn = 0.0;
    node_face(gj, hs_c, e_c, le_c, dx_c, hp_c, Pe_c, hs_n, e_n, le_n, dx_n, hp_n, Pe_n, D_c, k00, k10, k01, k11, Mn, PLn, tw, A_c,
              (YT && S == 1 && Eh) ? &hbn : nullptr);
    if constexpr (YT) {
      if (S == 1 && Eh) {  // emit the node north-east of an OUTPUT cell (every dual node belongs to exactly one thread)
        const bool own = lane >= FH && lane < FH + FOX && r0 + m >= FH && r0 + m <= (NR * TNW) - 1 - FH && gi <= g.nx - 2 && gj <= g.ny - 2;
        if (own) {
          const unsigned q = (unsigned)(gif + (g.nx - 1) * gj);
          const double wv = a.qw * tw;
          Eh[q] = hbn; Ev[q] = wv;
          emax[0] = fmax(emax[0], hbn); emax[1] = fmax(emax[1], fabs(wv));
        }
      }
    }
    if (S == 1 && th_red) {  // the node north-east of an OUTPUT cell belongs to this thread (every dual node to exactly one)
      const bool own = lane >= FH && lane < FH + FOX && r0 + m >= FH && r0 + m <= (NR * TNW) - 1 - FH && gi <= g.nx - 2 && gj <= g.ny - 2;
      thacc += own ? tw : 0.0;
      // dual-grid accumulator (gridded A): the node's own entry, owned by exactly this thread -- k_vjp_theta_strip<GACC>'s +=
      if constexpr (GA) { if (own) { const unsigned q = (unsigned)(gif + (g.nx - 1) * gj); Gp[q] = fma(a.qw, tw, Gp[q]); } }
    }
    // east face of this row, second term
    const double te = ((D_s + D_c) * g.hinv_dx2) * qe_c;
    const double Me = (dx_c < e_c.x && dx_c != -hs_c.x) ? te : 0.0;
    const double PLe = (dx_c > -hs_c.x && dx_c != e_c.x) ? -te : 0.0;
    const double W = dpp_from_west(k10 + PLe);  // what the lane to the west holds for this cell
    double v = ((k00 + W) + C_s) + (Me + Mn);
    v = hs_c.x > 0.0 ? v : 0.0;
    // 3S*+ stage update of lambda (k_adj_stage)
    const double dtk = dt * v;
    const double uo = u[m];
    double un;
    if (S == 1) {
      un = fma(bt, dtk, uo);
      if (ELDS) sEr[r0 + m][lane] = bh * dtk; else E[m] = bh * dtk;
    } else {
      const double t = fma(dl, uo, tmp[m]);
      un = fma(g1, uo, g2 * t);
      if (S >= 4) {
        if constexpr (RC) {
          un = fma(g3, rc->u0[m], un);
        } else {
          const int gjc = gj < 0 ? 0 : (gj > g.ny - 1 ? g.ny - 1 : gj);
          un = fma(g3, ldg32(src, (unsigned)(gif + g.nx * gjc)), un);
        }
      }
      un = fma(bt, dtk, un);
      if (dl != 0.0) tmp[m] = t;
      if (ELDS) sEr[r0 + m][lane] = fma(bh, dtk, sEr[r0 + m][lane]); else E[m] = fma(bh, dtk, E[m]);
    }
    u[m] = un;
    hs_c = hs_n; le_c = le_n; e_c = e_n; dx_c = dx_n; hp_c = hp_n; qe_c = qe_n; Pe_c = Pe_n;
    D_s = D_c; C_s = (k01 + dpp_from_west(k11)) + PLn;
    if constexpr (AF && ODINN_ADJ_APF) { A_c = A_n; asm volatile("" : "+v"(A_c)); }
    // row fence (see k_rk_fused_strip): pins the row order of this one-basic-block stage body
    if (ELDS) {
      if (S == 1)
        asm volatile("" : "+v"(u[m]), "+v"(hs_c.x), "+v"(hs_c.y), "+v"(le_c), "+v"(e_c.x), "+v"(e_c.y), "+v"(qe_c),
                     "+v"(Pe_c), "+v"(D_s), "+v"(C_s), "+v"(gif));
      else
        asm volatile("" : "+v"(u[m]), "+v"(tmp[m]), "+v"(hs_c.x), "+v"(hs_c.y), "+v"(le_c), "+v"(e_c.x), "+v"(e_c.y),
                     "+v"(qe_c), "+v"(Pe_c), "+v"(D_s), "+v"(C_s), "+v"(gif));
    } else if (S == 1)
      asm volatile("" : "+v"(u[m]), "+v"(E[m]), "+v"(hs_c.x), "+v"(hs_c.y), "+v"(le_c), "+v"(e_c.x), "+v"(e_c.y), "+v"(qe_c),
                   "+v"(Pe_c), "+v"(D_s), "+v"(C_s), "+v"(gif));
    else
      asm volatile("" : "+v"(u[m]), "+v"(E[m]), "+v"(tmp[m]), "+v"(hs_c.x), "+v"(hs_c.y), "+v"(le_c), "+v"(e_c.x), "+v"(e_c.y),
                   "+v"(qe_c), "+v"(Pe_c), "+v"(D_s), "+v"(C_s), "+v"(gif));
  }
  if (S == 1 && th_red) {  // reduced across the workgroup behind the barrier that ends the stage
    thacc = wave_sum(thacc);
    if (lane == 0) th_red[w] = thacc;
  }
  if (S < 5) {  // publish the strip's edge rows for the next stage: H at ITS time, lambda just updated
    const double swn = a.sitp[S < 5 ? S : 4];
    sE[wr][w][0][lane] = hs_itp(0, swn);
    sE[wr][w][1][lane] = hs_itp(NR - 1, swn);
    sLm[wr][w][0][lane] = lam_e(0);
    sLm[wr][w][1][lane] = lam_e(NR - 1);
    __syncthreads();
  }
}

// SKIP: exact ice-free shortcut (see below)
// GA (gridded A with a dual-grid accumulator): stage 1 of the step that follows a quadrature node also adds the node
// weights into A.Gacc (needs A.th_part)
constexpr bool adj_rc(bool AF, bool SG, int NR) { return ODINN_ADJ_RC && (AF || ODINN_ADJ_RC > 1) && SG && NR > 4; }
// The YT instantiation (Y law through its table) reads the glacier's table from global memory (L1 / L2-resident; two workgroups per
// CU, ~30 spilled registers at 128).  Measured against it and not adopted: -DODINN_ADJ_YT_LDS=1, the table in LDS (1024 intervals x
// 48 B = 48 KB next to the kernel's 76 KB, i.e. one workgroup per CU and 256 registers, three ds_read_b128 per node instead of three
// dependent L2 gathers) -- 174 -> 165 us per launch beside the contraction lanes at 8 x 512^2, 120.9 vs 121.2 ms per continuous
// gradient, 10.66 vs 10.68 gradient evaluations per second at 64 x 1024^2: the gathers are not what this kernel waits for;
// -DODINN_ADJ_YT_WPE=2 alone (256 registers, no spills): the same kernel time.
#define ODINN_ADJ_YT_LDS 0  // (fixed: its A/B is recorded above; no longer a build-time knob)
// (round 5, at 64 x 1024^2, ms per continuous gradient of 13 snapshots: 4 waves per SIMD with ~30 spilled registers 1036, the LDS
//  copy 938, 2 waves per SIMD and 256 registers without spills 903: the default now)
#ifndef ODINN_ADJ_YT_WPE
#define ODINN_ADJ_YT_WPE 2
#endif
constexpr int YT_LDS_NI = 1024;  // the table size the LDS copy is laid out for (odinn_batch::ytab_ni)
// SC -- the self-controlled reverse step: no controller / post-step launches (the forward solve's SC loop, sia2d_fused.hpp, for
// the reverse ODE).  Every workgroup of launch n first DECIDES attempt n - 1 of its glacier itself: wavefront 0 sums the glacier's
// error partials of launch n - 1 (controller_errsum: k_controller's order) and runs controller_decide -- the PID controller, the
// stop tables, the AdjState of the coming step -- on the state in A.gin / A.adj_in; all workgroups of a glacier compute the same
// decision from the same numbers, the one that owns tile (0, 0) publishes the new state to A.gout / A.adj_out (the two state arrays
// and the two partial arrays alternate between launches).  GState::pad bit 0 = an attempt awaits its decision.  When the decided
// step landed on a stop that changes lambda (a snapshot time: loss cotangent, mass-balance VJP, time-aggregated terms; a
// mass-balance-only stop), THIS launch is that glacier's post-step instead of an attempt: every workgroup applies adj_post_cell to
// its own output cells, lam[cur] -> lam[1 - cur], and the published state flips cur (pointwise: no halo, nobody reads what the
// launch writes); the next launch attempts the step the controller prepared.  A quadrature node needs nothing here: its theta-VJP
// is stage 1 of the next attempt (a.qw, A.th_part), as in the three-launch loop.
__device__ __forceinline__ double wave_uniform(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// UT: the U law (target :D) through its table -- D, alpha, beta of a node from the bi-quintic patch instead of the power law
// (256 registers: the patch's 36 coefficients do not fit 128)
template <bool AF, bool SKIP, bool SG = false, int NR = TRPT, bool GA = false, bool YT = false, bool SC = false, bool UT = false>
__global__ __launch_bounds__(TNT, (adj_rc(AF, SG, NR) ? 2 : ((YT || UT) ? ODINN_ADJ_YT_WPE : ODINN_FWPE))) void k_adj_fused_strip(Pools P, AdjFusedArgs A) {
  static_assert(!YT || (!AF && !GA), "the table's instantiation replaces the scalar A");
  static_assert(!UT || (!AF && !GA && !YT && !SC), "the U law's instantiation: its own law block, theta-VJP in launches of its own");
  constexpr bool RC = adj_rc(AF, SG, NR);
  __shared__ double2 sE[2][TNW][2][FRX];
  __shared__ double sLm[2][TNW][2][FRX];
  __shared__ double red[TNW];
  constexpr bool ELDS = ODINN_ADJ_ELDS && !(adj_rc(AF, SG, NR) && ODINN_ADJ_RC_EREG);
  __shared__ double sEr[ELDS ? (NR * TNW) : 1][FRX];
  const int4 t4 = A.tilesF[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * FOX - FH, gj0 = t4.z * (NR * TNW - 2 * FH) - FH;
  const int gi = gi0 + lane, r0 = NR * w;
  const bool inx = gi >= 0 && gi < g.nx, intx = gi >= 1 && gi <= g.nx - 2;
  const int gic = gi < 0 ? 0 : (gi > g.nx - 1 ? g.nx - 1 : gi);
  const int id0 = gi + g.nx * (gj0 + r0);
  const double* __restrict__ Bg = P.B + g.off;
  const double* __restrict__ Afg = AF ? P.Afield + g.offd : nullptr;
  auto idc = [&](int m) {  // the thread's cell of row m, clamped into the grid (what a clamped index picks up is never used)
    const int gj = gj0 + r0 + m;
    const int gjc = gj < 0 ? 0 : (gj > g.ny - 1 ? g.ny - 1 : gj);
    return (unsigned)(gic + g.nx * gjc);
  };
  AdjState a;
  double dt;
  int cur;
  if constexpr (SC) {
    static_assert(SG, "the self-controlled instantiations read the interleaved snapshot pairs");
    __shared__ GState s_gs;
    __shared__ AdjState s_ad;
    __shared__ int s_mode;  // 0: attempt a step, 1: this launch is the glacier's post-step, 2: the glacier is done
    if (threadIdx.x < 64) {
      GState sn = A.gin[t4.x];
      AdjState ad = A.adj_in[t4.x];
      // Everything the decision may read is requested at once, before the first use (one round trip through memory instead of
      // four dependent ones: state -> error partials -> stop tables -> snapshot times): the stop the attempt aimed at and the
      // one behind it, the snapshot times of the current segment.  (Rows past the glacier's table: clamped, never used.)
      CtrlPre pre;
      pre.is = sn.istop; pre.seg = ad.seg;
      {
        const int i0 = sn.istop < A.C.nrows ? sn.istop : A.C.nrows - 1, i1 = sn.istop + 1 < A.C.nrows ? sn.istop + 1 : A.C.nrows - 1;
        pre.n_stops = A.C.nstops[t4.x];
        pre.t_is = A.C.tstop(i0, t4.x); pre.t_is1 = A.C.tstop(i1, t4.x); pre.t_last = A.C.t_last;
        pre.mbf = A.C.at(A.C.mb_flag, i0, t4.x); pre.mbs = A.C.at(A.C.mb_slot, i0, t4.x);
        pre.snap = A.C.at(A.C.stop_snap, i0, t4.x); pre.hid = A.C.stop_hid ? A.C.at(A.C.stop_hid, i0, t4.x) : 0;
        pre.qw = A.C.stop_qw[(long long)i0 * A.C.G + t4.x];
        pre.ta = A.C.tsnap[(long long)ad.seg * A.C.G + t4.x]; pre.tb = A.C.tsnap[(long long)(ad.seg + 1) * A.C.G + t4.x];
      }
      double s, pw0, pw1, pw2;
      controller_errsum(A.C, g, sn.e2, sn.e3, lane, s, pw0, pw1, pw2);
      s = __shfl(s, 0, 64);
      int est = -1, mode = 0;
      bool newly_done = false;
      if (sn.done) {
        mode = 2;
      } else if (sn.pad & 1) {
        newly_done = controller_decide(sn, ad, g, A.C, t4.x, s, pw0, pw1, pw2, est, &pre) != 0;
        const bool post = sn.accepted && sn.at_stop && (ad.snapj >= 0 || (ad.pad > 0 && sn.mb_now && g.has_mb));
        mode = post ? 1 : (sn.done ? 2 : 0);
      }
      if (lane == 0) {
        s_gs = sn; s_ad = ad; s_mode = mode;
        if (t4.y == 0 && t4.z == 0) {  // the glacier's designated workgroup publishes the state for the next launch
          GState so = sn;
          if (mode == 1) so.cur = 1 - sn.cur;  // the post-step below leaves lambda in the other buffer
          so.pad = mode == 0 ? 1 : 0;
          A.gout[t4.x] = so;
          A.adj_out[t4.x] = ad;
          if (newly_done) atomicSub(A.C.n_active, 1);
          if (A.C.est_steps && est >= 0) A.C.est_steps[t4.x] = est;
          if (A.C.qw_out) A.C.qw_out[t4.x] = ad.qw;
        }
      }
    }
    __syncthreads();
    const int mode = __builtin_amdgcn_readfirstlane(s_mode);
    if (mode == 2) return;
    cur = __builtin_amdgcn_readfirstlane(s_gs.cur);
    dt = wave_uniform(s_gs.dt);
    a.seg = __builtin_amdgcn_readfirstlane(s_ad.seg); a.seg_stop = __builtin_amdgcn_readfirstlane(s_ad.seg_stop);
    a.snapj = __builtin_amdgcn_readfirstlane(s_ad.snapj); a.pad = __builtin_amdgcn_readfirstlane(s_ad.pad);
    a.qw = wave_uniform(s_ad.qw); a.s_stop = wave_uniform(s_ad.s_stop);
#pragma unroll
    for (int k = 0; k < 5; ++k) a.sitp[k] = wave_uniform(s_ad.sitp[k]);
    if (mode == 1) {
      const int mb_now = __builtin_amdgcn_readfirstlane(s_gs.mb_now), mb_slot = __builtin_amdgcn_readfirstlane(s_gs.mb_slot);
      const double* __restrict__ ps = (cur ? A.lam1 : A.lam0);
      double* __restrict__ pd = (cur ? A.lam0 : A.lam1);
      const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
      double wl = 0.0;
      long long roff = 0;
      if (a.snapj >= 0 && A.post.ws) {
        wl = A.post.ws[(long long)a.snapj * A.post.G + t4.x];
        if (wl != 0.0) roff = (long long)A.post.refslot[(long long)a.snapj * A.post.G + t4.x] * A.post.ntot;
      }
      if (lane >= FH && lane < FH + FOX && inx) {
#pragma unroll
        for (int m = 0; m < NR; ++m) {
          const int r = r0 + m, gj = gj0 + r;
          if (r >= FH && r <= (NR * TNW) - 1 - FH && gj < g.ny) {
            const long long id = g.off + gi + (long long)g.nx * gj;
            pd[id] = adj_post_cell(g, a, A.post, P.B, t4.x, mb_now, mb_slot, wl, roff, Ninv, id, ps[id]);
          }
        }
      }
      return;
    }
  } else {
    const GState* gs = P.gs + t4.x;
    if (gs->done) return;
    a = A.adj[t4.x];
    dt = gs->dt;
    cur = gs->cur;
  }
  // everything below is addressed relative to the glacier's first cell (block-uniform bases, 32-bit cell indices)
  const double* __restrict__ src = (cur ? A.lam1 : A.lam0) + g.off;
  double* __restrict__ dst = (cur ? A.lam0 : A.lam1) + g.off;
  const double* __restrict__ Ha = SG ? reinterpret_cast<const double*>(A.segs + (long long)a.seg * A.ntot + g.off)
                                     : A.snaps + (long long)a.seg * A.ntot + g.off;
  const double* __restrict__ Hb = Ha + A.ntot;  // (unused with SG)
  auto ld_ab = [&](unsigned id, double& ha, double& hb) {  // the two bracketing snapshots of a cell
    if constexpr (SG) {
      const double2 hd = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Ha) + ((size_t)id << 4));
      ha = hd.x; hb = hd.x + hd.y;
    } else {
      ha = ldg32(Ha, id); hb = ldg32(Hb, id);
    }
  };
  if (SKIP) {
    // Exact shortcut: if neither bracketing snapshot has ice anywhere on the halo region (and the five stage weights
    // lie in [0, 1], so that no interpolant has either), Hc = 0 on the region at every stage: every D and every
    // alpha, beta vanish and J_H^T lam = +0 on all cells, in all five stages.  The step then is the 3S*+ update of
    // lambda with a zero right-hand side -- the arithmetic below, bit-identical to running the stages.
    bool ice = false;
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      double ha, hb;
      ld_ab(idc(m), ha, hb);
      ice = ice || ha > 0.0 || hb > 0.0;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) ice = ice || !(a.sitp[k] >= 0.0 && a.sitp[k] <= 1.0);
    if (!__syncthreads_or(ice)) {
      const bool ocol = lane >= FH && lane < FH + FOX && inx;
      if constexpr (YT) {
        if (A.emitH && a.qw != 0.0) {  // an ice-free tile of a glacier on a quadrature node: its dual nodes carry (0, 0)
          double* __restrict__ eh = A.emitH + g.offd;
          double* __restrict__ ev = A.emitV + g.offd;
#pragma unroll
          for (int m = 0; m < NR; ++m) {
            const int r = r0 + m, gj = gj0 + r;
            if (ocol && r >= FH && r <= (NR * TNW) - 1 - FH && gi <= g.nx - 2 && gj <= g.ny - 2) {
              const unsigned q = (unsigned)(gi + (g.nx - 1) * gj);
              eh[q] = 0.0; ev[q] = 0.0;
            }
          }
        }
      }
      double errsq = 0.0;
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        const int r = r0 + m, gj = gj0 + r;
        if (r >= FH && r <= (NR * TNW) - 1 - FH && ocol && gj < g.ny) {
          const double up = ldg32(src, (unsigned)(id0 + g.nx * m));
          double un = fma(c_bt[0], 0.0, up), tm = up;  // stage 1 (tmp == u_n)
#pragma unroll
          for (int sg = 1; sg < 5; ++sg) {
            const double uo = un;
            const double t = fma(c_dl[sg], uo, tm);
            un = fma(c_g1[sg], uo, c_g2[sg] * t);
            if (sg >= 3) un = fma(c_g3[sg], up, un);
            un = fma(c_bt[sg], 0.0, un);
            if (c_dl[sg] != 0.0) tm = t;
          }
          stg32(dst, (unsigned)(id0 + g.nx * m), un);
          const double err = (un - up) - 0.0;
          const double sk = A.abstol + fmax(fabs(up), fabs(un)) * A.reltol;
          const double q = err / sk;
          errsq = fma(q, q, errsq);
        }
      }
      errsq = wave_sum(errsq);
      if (lane == 0) red[w] = errsq;
      __syncthreads();
      if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < TNW; ++k) sum += red[k];
        A.partF[t4.w] = sum;
      }
      return;
    }
  }
  double u[NR], tmp[NR], E[NR];
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int gj = gj0 + r0 + m;
    double l = 0.0;
    if (inx && gj >= 0 && gj < g.ny) l = ldg32(src, (unsigned)(id0 + g.nx * m));
    u[m] = l; tmp[m] = l; E[m] = 0.0;
  }
  {  // edge rows for stage 1
    auto edge = [&](int m, int e) {
      const int gj = gj0 + r0 + m;
      double h = 0.0, b = 0.0;
      if (inx && gj >= 0 && gj < g.ny) {
        const unsigned id = (unsigned)(id0 + g.nx * m);
        if constexpr (SG) {
          const double2 hd = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Ha) + ((size_t)id << 4));
          h = fma(a.sitp[0], hd.y, hd.x);
        } else {
          const double ha = ldg32(Ha, id);
          h = fma(a.sitp[0], ldg32(Hb, id) - ha, ha);
        }
        b = ldg32(Bg, id);
      }
      sE[0][w][e][lane] = cell_HS(h, b);
      sLm[0][w][e][lane] = (intx && gj >= 1 && gj <= g.ny - 2) ? u[m] : 0.0;
    };
    edge(0, 0);
    edge(NR - 1, 1);
  }
  [[maybe_unused]] AdjRowCache<RC ? NR : 1> rc;
  if constexpr (RC) {  // every load of the step is issued here (clamped indices: what they pick up outside the grid is never used)
    const bool nodex = gi >= 0 && gi <= g.nx - 2;
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      const unsigned id = idc(m);
      rc.hd[m] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Ha) + ((size_t)id << 4));
      rc.b[m] = ldg32(Bg, id);
      rc.u0[m] = ldg32(src, id);
    }
    if constexpr (AF) {
#pragma unroll
      for (int k = 0; k <= NR; ++k) {
        const int gj = gj0 + r0 + k - 1;
        const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
        rc.An[k] = ldg32(Afg, (unsigned)(ok ? gic + (g.nx - 1) * gj : 0));
      }
    }
  }
  constexpr bool YTL = YT && ODINN_ADJ_YT_LDS;
  __shared__ double2 sYt[YTL ? 3 * YT_LDS_NI : 1];
  if constexpr (YTL) {  // (the barrier below publishes it)
    const double2* __restrict__ tg = reinterpret_cast<const double2*>(A.ytab + g.yt_off);
    for (int k = threadIdx.x; k < 3 * YT_LDS_NI; k += TNT) sYt[k] = tg[k];
  }
  __syncthreads();
  __shared__ double th_red[TNW];
  // theta-VJP of the quadrature node the previous step reached (a.qw: its Gauss-Legendre weight, 0 otherwise; the controller
  // resets it at every call, so a repeated attempt after a rejection does not count the node twice)
  double* const thr = (A.th_part && a.qw != 0.0) ? th_red : nullptr;
  unsigned long long yt_beyond = 0ull;  // lanes whose node left the law table, any stage (raised at the kernel's end)
  const YtabRef yt{YTL ? reinterpret_cast<const double*>(sYt) : (YT ? A.ytab + g.yt_off : nullptr), A.ytab_over, A.ytab_ni, &yt_beyond};
  [[maybe_unused]] double emx[2] = {0.0, 0.0};
  [[maybe_unused]] const bool emit = YT && A.emitH != nullptr && a.qw != 0.0;
  [[maybe_unused]] LawDev Lu{};
  if constexpr (UT) {
    Lu.utab = A.utab; Lu.utab_nh = A.utab_nh; Lu.utab_ns = A.utab_ns; Lu.ut_inv_h = A.ut_inv_h; Lu.ut_inv_s = A.ut_inv_s;
    Lu.ytab_over = A.ytab_over;
  }
  adj_strip_stage<1, AF, SG, NR, GA, RC, YT, UT>(g, Afg, Ha, Hb, src, a, gic, gi, gj0, w, lane, dt, sE, sLm, u, tmp, E, Bg, sEr, thr,
                                                 GA ? A.Gacc + g.offd : nullptr, &rc, yt, emit ? A.emitH + g.offd : nullptr,
                                                 emit ? A.emitV + g.offd : nullptr, emx, &Lu);
  if constexpr (YT) {
    if (emit) {
      const double mh = wave_max(emx[0]), mv = wave_max(emx[1]);
      if (lane == 0) {
        if (mh > 0.0) atomicMax(A.emit_amax + t4.x, (unsigned long long)__double_as_longlong(mh));
        if (mv > 0.0) atomicMax(A.emit_vmax + t4.x, (unsigned long long)__double_as_longlong(mv));
      }
    }
  }
  if (thr && threadIdx.x == 0) {  // the tile's running sum, reduced per glacier once after the reverse solve
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < TNW; ++k) sum += th_red[k];
    A.th_part[t4.w] = fma(a.qw, sum, A.th_part[t4.w]);
  }
  adj_strip_stage<2, AF, SG, NR, false, RC, YT, UT>(g, Afg, Ha, Hb, src, a, gic, gi, gj0, w, lane, dt, sE, sLm, u, tmp, E, Bg, sEr, nullptr, nullptr, &rc, yt,
                                                     nullptr, nullptr, nullptr, &Lu);
  adj_strip_stage<3, AF, SG, NR, false, RC, YT, UT>(g, Afg, Ha, Hb, src, a, gic, gi, gj0, w, lane, dt, sE, sLm, u, tmp, E, Bg, sEr, nullptr, nullptr, &rc, yt,
                                                     nullptr, nullptr, nullptr, &Lu);
  adj_strip_stage<4, AF, SG, NR, false, RC, YT, UT>(g, Afg, Ha, Hb, src, a, gic, gi, gj0, w, lane, dt, sE, sLm, u, tmp, E, Bg, sEr, nullptr, nullptr, &rc, yt,
                                                     nullptr, nullptr, nullptr, &Lu);
  adj_strip_stage<5, AF, SG, NR, false, RC, YT, UT>(g, Afg, Ha, Hb, src, a, gic, gi, gj0, w, lane, dt, sE, sLm, u, tmp, E, Bg, sEr, nullptr, nullptr, &rc, yt,
                                                     nullptr, nullptr, nullptr, &Lu);
  // ---- output rows [FH, (NR * TNW)-1-FH]: lam' from the registers, embedded error partial -----------------------
  const bool ocol = lane >= FH && lane < FH + FOX && inx;
  double errsq = 0.0;
  double upf[NR];
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int r = r0 + m, gj = gj0 + r;
    const bool out = r >= FH && r <= (NR * TNW) - 1 - FH && ocol && gj < g.ny;
    if constexpr (RC) upf[m] = rc.u0[m];
    else upf[m] = ldg32(src, (unsigned)(out ? id0 + g.nx * m : 0));
  }
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int r = r0 + m, gj = gj0 + r;
    if (r >= FH && r <= (NR * TNW) - 1 - FH && ocol && gj < g.ny) {
      const double upv = upf[m];
      stg32(dst, (unsigned)(id0 + g.nx * m), u[m]);
      const double err = (u[m] - upv) - (ELDS ? sEr[r0 + m][lane] : E[m]);
      const double sk = A.abstol + fmax(fabs(upv), fabs(u[m])) * A.reltol;
      const double q = err / sk;
      errsq = fma(q, q, errsq);
    }
  }
  errsq = wave_sum(errsq);
  if (lane == 0) red[w] = errsq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < TNW; ++k) sum += red[k];
    A.partF[t4.w] = sum;
  }
}

// ================= H-VJP in the strip layout (integer-power A law, DiscreteVJP) ===================================
// VJP_lambda_dSIA/dH_discrete (adjoint.jl:99-148) -- k_vjp_H's contract for law mode 0 -- with the face form and the
// register / DPP layout of adj_strip_stage and the geometry of k_dhdt_strip: a wavefront owns DNR contiguous rows of a
// 64-wide region (62 x 62 outputs, one-cell halo), every load of the thread (H, B, lambda, the A nodes, the loss data) is
// in flight before the first use, the first / last row of a strip cross wavefronts through LDS once.
//   MODE 0: out = J_H^T lam;  MODE 1 (one reverse-Euler step of the DiscreteAdjoint, gradient.jl:235-242):
//   out = lam + dt_g J_H^T lam + w_g (2 / N) mask (H - Href), loss partial of the tile into its glacier's partial slot.
// Tiles come from the RHS strip table (tilesD); partials go to the glacier's slots of the regular tile table
// (tile0 + local index; the slots past ntilesD are zeroed by the glacier's first tile) so that every reduction over
// P.part keeps working unchanged.
// ODINN_VJPH_RC: the reverse-Euler form (MODE 1) runs at 2 wavefronts per SIMD with EVERY load of the thread -- bed, A nodes,
// reference thickness, mask -- issued before the sweep, like MODE 0's H and lambda.  At 128 registers those were fetched one
// row ahead inside the sweep, i.e. each row waited for its own loads: 0.74 ms per 64 x 1024^2 against 0.36 ms for MODE 0,
// which moves 32 of MODE 1's 41 B/cell.
#define ODINN_VJPH_RC 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
// YT: the Y law through its table with Y(Hbar) in A's place plus the reference's forward-difference term of dD/dHbar (as in
// adj_strip_stage<..., YT>; 256 registers: the table evaluation does not fit 128 without spills)
template <bool AF, int MODE, bool YT = false>
__global__ __launch_bounds__(TNT, (((ODINN_VJPH_RC && MODE == 1) || YT) ? 2 : ODINN_FWPE)) void k_vjp_H_strip(Pools P, const int4* __restrict__ tilesD, AdjArgs A) {
  static_assert(!YT || !AF, "the table replaces the scalar A");
  constexpr bool RCV = ODINN_VJPH_RC && MODE == 1;
  __shared__ double2 sE[TNW][2][FRX];
  __shared__ double sLm[TNW][2][FRX];
  __shared__ double red[TNW];
  const int4 t4 = tilesD[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * DOX - 1, gj0 = t4.z * DOY - 1;
  const int gi = gi0 + lane, r0 = DNR * w;
  const bool inx = gi >= 0 && gi < g.nx, intx = gi >= 1 && gi <= g.nx - 2;
  const bool ocol = lane >= 1 && lane <= DOX && inx;
  [[maybe_unused]] unsigned long long yt_beyond = 0ull;  // (YT) lanes whose node left the law table: raised at the kernel's end (ytab_eval_acc)
  const int id0 = gi + g.nx * (gj0 + r0);
  const double* __restrict__ Hg = A.H + g.off;
  const double* __restrict__ Bg = P.B + g.off;
  const double* __restrict__ Lg = A.lam + g.off;
  double* __restrict__ dst = A.out + g.off;
  const long long slot = 4 * (long long)(g.tile0 + (t4.w - g.tile0D)) + 1;
  // H and lambda of the thread's rows are all in flight before the first use; the bed (and A, and the loss data) of a row
  // is fetched one row ahead of its use inside the sweep -- with all of them resident the kernel does not fit 128 VGPRs
  double hh[DNR], ll[DNR];
#pragma unroll
  for (int m = 0; m < DNR; ++m) {
    const int gj = gj0 + r0 + m;
    const bool ok = inx && gj >= 0 && gj < g.ny;
    hh[m] = ok ? ldg32(Hg, (unsigned)(id0 + g.nx * m)) : 0.0;
    ll[m] = ok ? ldg32(Lg, (unsigned)(id0 + g.nx * m)) : 0.0;
  }
  int idf = id0;
  [[maybe_unused]] double bbr[RCV ? DNR : 1];
  if constexpr (RCV) {
#pragma unroll
    for (int m = 0; m < DNR; ++m) {
      const int gj = gj0 + r0 + m;
      const bool ok = inx && gj >= 0 && gj < g.ny;
      bbr[m] = ok ? ldg32(Bg, (unsigned)(id0 + g.nx * m)) : 0.0;
    }
  }
  auto bed = [&](int m) {
    if constexpr (RCV) return bbr[m];
    const int gj = gj0 + r0 + m;
    const bool ok = inx && gj >= 0 && gj < g.ny;
    return ok ? ldg32(Bg, (unsigned)(idf + g.nx * m)) : 0.0;
  };
  const double b_first = bed(0), b_last = bed(DNR - 1);
  // A of the node row gj0 + r0 - 1 + k of the thread's column (0 outside the dual grid: such nodes only feed masked cells);
  // fetched one row ahead of its use inside the sweep, like the loss data, to keep the kernel inside 128 VGPRs
  const double* __restrict__ Afg = AF ? P.Afield + g.offd : nullptr;
  const bool nodex = gi >= 0 && gi <= g.nx - 2;
  // idf: the thread's first cell index again, but opaque after every row fence of the sweep -- keeps the loads a row issues
  // (A of the next node row, loss data and lambda of the next row) inside that row instead of all at the top
  [[maybe_unused]] double anr[(RCV && AF) ? DNR + 1 : 1];
  if constexpr (RCV && AF) {
#pragma unroll
    for (int k = 0; k <= DNR; ++k) {
      const int gj = gj0 + r0 - 1 + k;
      const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
      anr[k] = ok ? ldg32(Afg, (unsigned)(gi + (g.nx - 1) * gj)) : 0.0;
    }
  }
  auto a_node = [&](int k) {
    if constexpr (RCV && AF) return anr[k];
    const int gj = gj0 + r0 - 1 + k;
    const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
    return ok ? ldg32(Afg, (unsigned)((idf - id0) + gi + (g.nx - 1) * gj)) : 0.0;
  };
  // loss data of the thread's output cells (MODE 1 at a data stop): H - Href where the mask is set
  double dt = 1.0, wl = 0.0;
  // (without reference data at this stop the two pointers fall back to H: the loads stay unconditional, their values unused)
  const unsigned char* __restrict__ Mg = reinterpret_cast<const unsigned char*>(Hg);
  const double* __restrict__ Rg = Hg;
  if (MODE == 1) {
    dt = A.dts[t4.x];
    wl = A.ws ? A.ws[t4.x] : 0.0;
    if (wl != 0.0) {
      const long long roff = (long long)A.refslot[t4.x] * A.ntot + g.off;
      Mg = A.mask + roff; Rg = A.Href + roff;
    }
    // the partial slots of the regular table that the strip tiles of this glacier do not use
    if (t4.w == g.tile0D)
      for (int k = g.ntilesD + threadIdx.x; k < g.ntiles; k += TNT) P.part[4 * (long long)(g.tile0 + k) + 1] = 0.0;
  }
  auto lam_e = [&](int m) {  // lambda masked to the interior cells (the transposed difference of adjoint.jl:100-101)
    const int gj = gj0 + r0 + m;
    return (intx && gj >= 1 && gj <= g.ny - 2) ? ll[m] : 0.0;
  };
  [[maybe_unused]] double hrr[RCV ? DNR : 1];
  [[maybe_unused]] unsigned mkr[RCV ? DNR : 1];
  if constexpr (RCV) {  // (clamped to the thread's first cell where it has no output cell: never used there)
#pragma unroll
    for (int m = 0; m < DNR; ++m) {
      const int r = r0 + m, gj = gj0 + r;
      const bool oc = ocol && r >= 1 && r <= DOY && gj < g.ny;
      const unsigned id = (unsigned)(oc ? id0 + g.nx * m : (inx && gj0 + r0 >= 0 && gj0 + r0 < g.ny ? id0 : 0));
      mkr[m] = Mg[id];
      hrr[m] = ldg32(Rg, id);
    }
  }
  bool nz = false;
#pragma unroll
  for (int m = 0; m < DNR; ++m) nz = nz || (hh[m] > 0.0);
  // dt = 0: the glacier has no stop in this row of the per-glacier stop tables (see k_vjp_H) -- the ice-free branch below
  // then passes lambda through (wl = 0 there as well) whatever the unused snapshot slot holds
  if (MODE == 1 && dt == 0.0) nz = false;
  sE[w][0][lane] = cell_HS(hh[0], b_first);
  sE[w][1][lane] = cell_HS(hh[DNR - 1], b_last);
  sLm[w][0][lane] = lam_e(0);
  sLm[w][1][lane] = lam_e(DNR - 1);
  const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
  double lsum = 0.0;
  // no ice on the whole region: the result is masked by the cell's own H > 0 (adjoint.jl:148) -> J^T lam = 0 exactly
  const bool ice = __syncthreads_or(nz);
  if (ice) {
    const double Gq = g.Gam * (1.0 / 1024.0);
    const int wb = w > 0 ? w - 1 : 0, eb = w > 0 ? 1 : 0, wt = w + 1 < TNW ? w + 1 : w, et = w + 1 < TNW ? 0 : 1;
    const double2 hs_s = sE[wb][eb][lane], hs_top = sE[wt][et][lane];
    const double le_s = sLm[wb][eb][lane], le_top = sLm[wt][et][lane];
    double2 hs_c = cell_HS(hh[0], b_first);
    double le_c = lam_e(0);
    double2 e_c = dpp_from_east(hs_c);
    double dx_c = e_c.y - hs_c.y, hp_c = hs_c.x + e_c.x, qe_c = dpp_shift(le_c, false) - le_c;
    double Pe_c = qe_c * clampn(dx_c, e_c.x, hs_c.x);
    double D_s, C_s;
    double b_nx = bed(DNR > 1 ? 1 : 0);  // bed of row m + 1
    // of row m, if it is an output cell: H - Href where the mask is set, and lambda of the cell AGAIN (an L2 hit; keeping the
    // first copy alive until the row's result exists costs 16 VGPRs, i.e. a spill to scratch)
    auto loss_data = [&](int m, double& d, bool& on, double& lo) {
      d = 0.0; on = false; lo = 0.0;
      if (MODE == 1) {
        const int r = r0 + m, gj = gj0 + r;
        const bool oc = ocol && r >= 1 && r <= DOY && gj < g.ny;
        if constexpr (RCV) {
          if (oc) {
            lo = ll[m];
            on = wl != 0.0 && mkr[m] != 0;
            d = on ? hh[m] - hrr[m] : 0.0;
          }
        } else if (oc) {
          const unsigned id = (unsigned)(idf + g.nx * m);
          const unsigned char mk = Mg[id];
          const double hr = ldg32(Rg, id);
          lo = ldg32(Lg, id);
          on = wl != 0.0 && mk != 0;
          d = on ? hh[m] - hr : 0.0;
        }
      }
    };
    auto node_face = [&](double an, double2 hs_lo, double2 e_lo, double le_lo, double dx_lo, double hp_lo, double Pe_lo,
                         double2 hs_hi, double2 e_hi, double le_hi, double dx_hi, double hp_hi, double Pe_hi, double& D,
                         double& k00, double& k10, double& k01, double& k11, double& Mn, double& PLn) {
      const double dyw = hs_hi.y - hs_lo.y, dye = e_hi.y - e_lo.y;
      const double qn = le_hi - le_lo;
      const double Pn = qn * clampn(dyw, hs_hi.x, hs_lo.x);
      const double Pn_e = dpp_shift(Pn, false);
      const double gx = (dx_lo + dx_hi) * g.hinv_dx, gy = (dyw + dye) * g.hinv_dy;
      const double Hs = hp_lo + hp_hi;  // 4 Hbar
      const double gS2 = gx * gx + gy * gy;
      double An = AF ? an : g.A;
      [[maybe_unused]] double Yp = 0.0;
      if constexpr (YT) An = ytab_eval_acc<true>(A.ytab + g.yt_off, A.ytab_ni, yt_beyond, g.yt_inv_h, 0.25 * Hs, Yp);
      const double Kq = An * Gq;
      const double H2 = Hs * Hs, H4 = H2 * H2, H5 = H4 * Hs;
      D = (Kq * H5) * gS2;
      const double Da = -fma(g.hinv_dx2, Pe_lo + Pe_hi, g.hinv_dy2 * (Pn + Pn_e));
      double ad = (((Kq * 5.0) * H4) * gS2) * Da;  // alpha Da / 4
      if constexpr (YT) {
        const double geo = (Gq * H5) * gS2;  // Gam Hbar^5 |grad S|^2
        ad = fma(0.25 * ((Yp * geo - An * geo) * (1.0 / 1e-4)), Da, ad);  // (target_D_hybrid.jl:58-71; see adj_strip_stage)
      }
      const double bd = ((Kq * 2.0) * H5) * Da;           // beta Da
      const double bx = g.hinv_dx * (bd * gx), by = g.hinv_dy * (bd * gy);
      const double am = ad - bx, ap = ad + bx;
      k00 = am - by; k10 = ap - by; k01 = am + by; k11 = ap + by;
      const double Dw = dpp_from_west(D);
      const double tn = ((Dw + D) * g.hinv_dy2) * qn;
      Mn = (dyw < hs_hi.x && dyw != -hs_lo.x) ? tn : 0.0;
      PLn = (dyw > -hs_lo.x && dyw != hs_hi.x) ? -tn : 0.0;
    };
    {  // the node row and the north faces just below the strip
      const double2 e_s = dpp_from_east(hs_s);
      const double lee_s = dpp_shift(le_s, false);
      const double dx_s = e_s.y - hs_s.y, hp_s = hs_s.x + e_s.x;
      const double Pe_s = (lee_s - le_s) * clampn(dx_s, e_s.x, hs_s.x);
      double k00, k10, k01, k11, Mn, PLn;
      node_face(AF ? a_node(0) : 0.0, hs_s, e_s, le_s, dx_s, hp_s, Pe_s, hs_c, e_c, le_c, dx_c, hp_c, Pe_c, D_s, k00, k10, k01, k11, Mn, PLn);
      C_s = (k01 + dpp_from_west(k11)) + PLn;
    }
    double a_nx = AF ? a_node(1) : 0.0, hd_nx, lo_nx;
    bool hm_nx;
    loss_data(0, hd_nx, hm_nx, lo_nx);
#pragma unroll
    for (int m = 0; m < DNR; ++m) {
      const int r = r0 + m, gj = gj0 + r;
      const double a_c = a_nx, hd_c = hd_nx, lo_c = lo_nx;
      const bool hm_c = hm_nx;
      if (m + 1 < DNR) {
        if (AF) a_nx = a_node(m + 2 <= DNR ? m + 2 : m);
        loss_data(m + 1 < DNR ? m + 1 : m, hd_nx, hm_nx, lo_nx);
      }
      const double2 hs_n = m + 1 < DNR ? cell_HS(hh[m + 1 < DNR ? m + 1 : m], b_nx) : hs_top;
      if (m + 2 < DNR) b_nx = m + 2 == DNR - 1 ? b_last : bed(m + 2 < DNR ? m + 2 : m);
      const double le_n = m + 1 < DNR ? lam_e(m + 1 < DNR ? m + 1 : m) : le_top;
      const double2 e_n = dpp_from_east(hs_n);
      const double lee_n = dpp_shift(le_n, false);
      const double dx_n = e_n.y - hs_n.y, hp_n = hs_n.x + e_n.x, qe_n = lee_n - le_n;
      const double Pe_n = qe_n * clampn(dx_n, e_n.x, hs_n.x);
      double D_c, k00, k10, k01, k11, Mn, PLn;
      node_face(a_c, hs_c, e_c, le_c, dx_c, hp_c, Pe_c, hs_n, e_n, le_n, dx_n, hp_n, Pe_n, D_c, k00, k10, k01, k11, Mn, PLn);
      const double te = ((D_s + D_c) * g.hinv_dx2) * qe_c;
      const double Me = (dx_c < e_c.x && dx_c != -hs_c.x) ? te : 0.0;
      const double PLe = (dx_c > -hs_c.x && dx_c != e_c.x) ? -te : 0.0;
      const double W = dpp_from_west(k10 + PLe);
      double v = ((k00 + W) + C_s) + (Me + Mn);
      v = hs_c.x > 0.0 ? v : 0.0;  // dlam .* (H .> 0)  (adjoint.jl:148)
      if (ocol && r >= 1 && r <= DOY && gj < g.ny) {
        double o = v;
        if (MODE == 1) {
          o = fma(dt, v, lo_c);
          if (hm_c) {
            o = fma(wl * 2.0 * Ninv, hd_c, o);
            lsum = fma(hd_c, hd_c, lsum);
          }
        }
        stg32(dst, (unsigned)(id0 + g.nx * m), o);
      }
      hs_c = hs_n; le_c = le_n; e_c = e_n; dx_c = dx_n; hp_c = hp_n; qe_c = qe_n; Pe_c = Pe_n;
      D_s = D_c; C_s = (k01 + dpp_from_west(k11)) + PLn;
      asm volatile("" : "+v"(hs_c.x), "+v"(hs_c.y), "+v"(le_c), "+v"(e_c.x), "+v"(e_c.y), "+v"(qe_c), "+v"(Pe_c), "+v"(D_s),
                   "+v"(C_s), "+v"(idf));
    }
  } else {
#pragma unroll
    for (int m = 0; m < DNR; ++m) {
      const int r = r0 + m, gj = gj0 + r;
      if (ocol && r >= 1 && r <= DOY && gj < g.ny) {
        double o = 0.0;
        if (MODE == 1) {
          o = fma(dt, 0.0, ll[m]);
          if (wl != 0.0 && (RCV ? mkr[m] != 0 : Mg[(unsigned)(id0 + g.nx * m)] != 0)) {
            const double hd = hh[m] - (RCV ? hrr[m] : ldg32(Rg, (unsigned)(id0 + g.nx * m)));
            o = fma(wl * 2.0 * Ninv, hd, o);
            lsum = fma(hd, hd, lsum);
          }
        }
        stg32(dst, (unsigned)(id0 + g.nx * m), o);
      }
    }
  }
  if (MODE == 1) {
    lsum = wave_sum(lsum);
    if (lane == 0) red[w] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < TNW; ++k) sum += red[k];
      P.part[slot] = sum * wl * Ninv;
    }
  }
  if constexpr (YT) ytab_raise(A.ytab_over, yt_beyond);
}

// ================= theta-VJP in the strip layout (integer-power A-type laws) ========================================
// VJP_lambda_dSIA/dtheta_discrete (adjoint.jl:178-255) for the laws whose dD/dtheta factors as (dA/dtheta) x spat(H):
// the reduction  sum_nodes scale_g spat D_adjoint  -- k_vjp_theta<LM_FAST>'s contract (tile partial in slot 2, optional
// dual-grid accumulator Gacc, per-glacier scale, in-place accumulation, lambda from the per-glacier ping-pong buffer, H
// formed from two bracketing snapshots at the stop of the reverse solve) -- with the layout of k_vjp_H_strip.  A thread
// owns the node at the north-east corner of each of its output cells; D_adjoint of that node comes from the east faces of
// the row and the row above and the north faces of the cell and its east neighbour (the face form of adj_strip_stage).
// EMIT (Y law through its table, `:Linear` gradient of the law): instead of reducing, every owned node writes (Hbar, weight) into
// A.emitH / A.emitV (pre-zeroed by the caller: tiles that leave early contribute zeros) -- k_vjp_theta's emit mode for the
// integer-power form of the Y law's geometry factor (GDev::yt_fast)
template <bool GACC, bool ITP, bool EMIT = false>
__global__ __launch_bounds__(TNT, ODINN_FWPE) void k_vjp_theta_strip(Pools P, const int4* __restrict__ tilesD, ThArgs A) {
  __shared__ double2 sE[TNW][FRX];
  __shared__ double sLm[TNW][FRX];
  __shared__ double red[TNW];
  const int4 t4 = tilesD[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const long long slot = 4 * (long long)(g.tile0 + (t4.w - g.tile0D)) + 2;
  const double scale = A.scales ? A.scales[t4.x] : 1.0;
  // the partial slots of the regular table that this glacier's strip tiles do not use (in-place accumulation: the caller
  // zeroed the whole table once)
  if (!EMIT && !A.accum && t4.w == g.tile0D)
    for (int k = g.ntilesD + threadIdx.x; k < g.ntiles; k += TNT) P.part[4 * (long long)(g.tile0 + k) + 2] = 0.0;
  if (scale == 0.0) {  // this glacier contributes nothing now (e.g. not at a quadrature node)
    if (!EMIT && threadIdx.x == 0 && !A.accum) P.part[slot] = 0.0;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * DOX - 1, gj0 = t4.z * DOY - 1;
  const int gi = gi0 + lane, r0 = DNR * w;
  const bool inx = gi >= 0 && gi < g.nx, intx = gi >= 1 && gi <= g.nx - 2;
  const bool ocol = lane >= 1 && lane <= DOX && inx;
  const int id0 = gi + g.nx * (gj0 + r0);
  const double* __restrict__ Bg = P.B + g.off;
  const double* __restrict__ Lg = ((A.lam_alt && P.gs[t4.x].cur) ? A.lam_alt : A.lam) + g.off;
  const double* __restrict__ Hg = A.H + g.off;
  [[maybe_unused]] const double* __restrict__ Hb = nullptr;
  [[maybe_unused]] double sw = 0.0;
  if (ITP) {
    const AdjState a = A.adj[t4.x];
    Hg = A.snaps + (long long)a.seg_stop * A.ntot + g.off;
    Hb = Hg + A.ntot;
    sw = a.s_stop;
  }
  double2 hs[DNR];
  double le[DNR];
  bool nz = false;
#pragma unroll
  for (int m = 0; m < DNR; ++m) {
    const int gj = gj0 + r0 + m;
    const bool ok = inx && gj >= 0 && gj < g.ny;
    double h = ok ? ldg32(Hg, (unsigned)(id0 + g.nx * m)) : 0.0;
    if (ITP) {
      const double hb = ok ? ldg32(Hb, (unsigned)(id0 + g.nx * m)) : 0.0;
      h = fma(sw, hb - h, h);  // load_tile_HS2's formula
    }
    const double bb = ok ? ldg32(Bg, (unsigned)(id0 + g.nx * m)) : 0.0;
    const double l = ok ? ldg32(Lg, (unsigned)(id0 + g.nx * m)) : 0.0;
    hs[m] = cell_HS(h, bb);
    le[m] = (intx && gj >= 1 && gj <= g.ny - 2) ? l : 0.0;  // lambda masked to the interior
    nz = nz || h > 0.0;
  }
  sE[w][lane] = hs[0];
  sLm[w][lane] = le[0];
  // no ice anywhere on the region: every owned node has Hbar = 0, its weight vanishes identically (k_vjp_theta's shortcut)
  if (!__syncthreads_or(nz)) {
    if (!EMIT && threadIdx.x == 0 && !A.accum) P.part[slot] = 0.0;
    return;
  }
  const double Gq = g.Gam * (1.0 / 1024.0);
  const double2 hs_top = sE[w + 1 < TNW ? w + 1 : w][lane];  // (the last wavefront's rows DNR-1 are never output rows)
  const double le_top = sLm[w + 1 < TNW ? w + 1 : w][lane];
  double2 hs_c = hs[0];
  double le_c = le[0];
  double2 e_c = dpp_from_east(hs_c);
  double dx_c = e_c.y - hs_c.y, hp_c = hs_c.x + e_c.x;
  double Pe_c = (dpp_shift(le_c, false) - le_c) * clampn(dx_c, e_c.x, hs_c.x);
  double acc = 0.0;
#pragma unroll
  for (int m = 0; m < DNR; ++m) {
    const int r = r0 + m, gj = gj0 + r;
    const double2 hs_n = m + 1 < DNR ? hs[m + 1 < DNR ? m + 1 : m] : hs_top;
    const double le_n = m + 1 < DNR ? le[m + 1 < DNR ? m + 1 : m] : le_top;
    const double2 e_n = dpp_from_east(hs_n);
    const double dx_n = e_n.y - hs_n.y, hp_n = hs_n.x + e_n.x;
    const double Pe_n = (dpp_shift(le_n, false) - le_n) * clampn(dx_n, e_n.x, hs_n.x);
    const double dyw = hs_n.y - hs_c.y, dye = e_n.y - e_c.y;
    const double Pn = (le_n - le_c) * clampn(dyw, hs_n.x, hs_c.x);
    const double Pn_e = dpp_shift(Pn, false);
    const double gx = (dx_c + dx_n) * g.hinv_dx, gy = (dyw + dye) * g.hinv_dy;
    const double Hs = hp_c + hp_n;  // 4 Hbar
    const double gS2 = gx * gx + gy * gy;
    const double H2 = Hs * Hs, H4 = H2 * H2;
    const double Da = -fma(g.hinv_dx2, Pe_c + Pe_n, g.hinv_dy2 * (Pn + Pn_e));
    const double wgt = scale * (((Gq * (H4 * Hs)) * gS2) * Da);
    if (ocol && r >= 1 && r <= DOY && gi <= g.nx - 2 && gj <= g.ny - 2) {  // gj >= 0 and gi >= 0 hold for output cells
      if constexpr (EMIT) {
        const long long q = g.offd + gi + (long long)(g.nx - 1) * gj;
        A.emitH[q] = 0.25 * Hs;
        A.emitV[q] = wgt;
      } else {
        acc += wgt;
        if (GACC) A.Gacc[g.offd + gi + (long long)(g.nx - 1) * gj] += wgt;
      }
    }
    hs_c = hs_n; le_c = le_n; e_c = e_n; dx_c = dx_n; hp_c = hp_n; Pe_c = Pe_n;
  }
  if constexpr (EMIT) return;
  acc = wave_sum(acc);
  if (lane == 0) red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < TNW; ++k) sum += red[k];
    P.part[slot] = A.accum ? P.part[slot] + sum : sum;
  }
}

}  // namespace odinn
