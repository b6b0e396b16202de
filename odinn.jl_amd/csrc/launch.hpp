// launch.hpp -- host-callable launch wrappers; each law mode (LM) of the stencil kernels is
// compiled in its own translation unit (k_fwd.hip / k_adj.hip / k_fused.hip with -DODINN_LM=0 ... 8).
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

#define ODINN_DECL_LM(LM)                                                                                      \
  void launch_dhdt_lm##LM(int nblk, hipStream_t st, Pools P, LawDev L, const double* U, double* dH, int base); \
  void launch_euler_cfl_lm##LM(int nblk, hipStream_t st, Pools P, LawDev L, const double* src, double* dst);   \
  void launch_rk_stage_lm##LM(int stage, int nblk, hipStream_t st, Pools P, LawDev L, const double* src,       \
                              double* dst, double* S2, double* S3, double* E, double abstol, double reltol);   \
  void launch_vjp_H_lm##LM(int mode, int vj, int nblk, hipStream_t st, Pools P, LawDev L, AdjArgs A, int base); \
  void launch_vjp_theta_lm##LM(int nblk, hipStream_t st, Pools P, LawDev L, ThArgs A, int base);           \
  void launch_adj_stage_lm##LM(int stage, int vj, int nblk, hipStream_t st, Pools P, LawDev L, AdjStageArgs A); \
  void launch_rk_fused_lm##LM(int nblk, hipStream_t st, Pools P, LawDev L, const int4* tilesF, double* U0,     \
                              double* U1, double* partF, double abstol, double reltol, int skip, int small);
ODINN_DECL_LM(0)
ODINN_DECL_LM(1)
ODINN_DECL_LM(2)
ODINN_DECL_LM(3)
ODINN_DECL_LM(4)
ODINN_DECL_LM(5)
ODINN_DECL_LM(6)
ODINN_DECL_LM(7)  // the Y law through its per-glacier table (LM_YTAB)
ODINN_DECL_LM(8)  // the U law through the batch's bivariate table (LM_UTAB)
#undef ODINN_DECL_LM

// k_fused.hip, law mode 0 only
void launch_rk_fused_strip(int nblk, int afield, int rows, hipStream_t st, Pools P, LawDev L, const int4* tilesF, double* U0,
                           double* U1, double* partF, double abstol, double reltol, int skip, const ScArgs* sc = nullptr, int sq = 0,
                           int ytab = 0);

void launch_dhdt_strip(int nblk, int afield, int skip, hipStream_t st, Pools P, const int4* tilesD, const double* U, double* dH);
void launch_euler_cfl_strip(int nblk, int afield, hipStream_t st, Pools P, const int4* tilesD, const double* src, double* dst, double* partD);
constexpr int DHDT_OX = 62, DHDT_OY = 62;  // output tile of k_dhdt_strip (sia2d_fused.hpp: DOX, DOY)

// k_adjf.hip, law mode 0 only
void launch_adj_fused_strip(int nblk, int afield, int skip, int rows, hipStream_t st, Pools P, AdjFusedArgs A, int sc = 0);
void launch_adj_fused_lds(int nblk, int skip, hipStream_t st, Pools P, LawDev L, AdjFusedArgs A);  // k_adjl.hip
void launch_vjp_H_strip(int mode, int afield, int nblk, hipStream_t st, Pools P, const int4* tilesD, AdjArgs A);
void launch_vjp_theta_strip(int gacc, int itp, int nblk, hipStream_t st, Pools P, const int4* tilesD, ThArgs A);

// k_interp.hip: `:Linear` spatial interpolation of d law / d theta for the Y law (target_D_hybrid.jl:136-160)
constexpr int INTERP_KMAX = 512;
// odinn_batch::d_ucell per lane (8-byte words): 4 (KMAX - 1)^2 corner sums, 8 (KMAX - 1)^2 fixed-point limbs, max |v| (launch_interp_theta_U)
constexpr size_t INTERP_UCELL_WORDS = (size_t)12 * (INTERP_KMAX - 1) * (INTERP_KMAX - 1) + 8;
size_t interp_batch_temp_bytes(long long n_max);
size_t interp_batch_lds_bytes(int P);
size_t node_backprop_part_count(int ng, int Pn);
int launch_node_backprop(hipStream_t st, Pools P, const LawDev& L, int g0, int ng, long long end_all, const double* nodeH,
                         const double* nodeS, const double* nodeV, double* part, double* dth, int accumulate);
void launch_fill_gid(hipStream_t st, Pools P, int G, long long ntotd, unsigned* gid, unsigned* iota);
int launch_interp_theta_batch(hipStream_t st, Pools P, const LawDev& L, int n_half, int g0, int ng, long long lo, long long n,
                              const double* nodeH, const double* nodeV, const unsigned* gid, const unsigned* iota, double* sH,
                              double* sV, unsigned* iA, unsigned* iB, unsigned* kA, unsigned* kB, void* tmp, size_t tmp_bytes,
                              double* knots, int* M, double* ab, double* dth, int accumulate);
size_t interp_active_temp_bytes(long long n);
size_t interp_select_scratch_bytes(int G, size_t* zero_bytes);
int launch_interp_theta_select(hipStream_t st, Pools P, const LawDev& L, int n_half, int G, long long n_act, const double* nodeH,
                               const double* nodeV, const unsigned* act, const long long* aoff, double* buf, void* scratch,
                               double* knots, int* M, double* ab, double* dth, int accumulate,
                               const unsigned long long* amax_in = nullptr, const unsigned long long* vmax_in = nullptr);
int launch_interp_active(hipStream_t st, Pools P, int G, long long ntotd, const unsigned* gid, const unsigned* iota, const double* snaps,
                         int nslots, long long ntot, unsigned char* flags, void* tmp, size_t tmp_bytes, unsigned* act, unsigned* gid_act,
                         long long* aoff, unsigned* n_act_dev);
int launch_interp_theta_active(hipStream_t st, Pools P, const LawDev& L, int n_half, int G, long long n_act, const double* nodeH,
                               const double* nodeV, const unsigned* act, const unsigned* gid_act, const long long* aoff,
                               const unsigned* iota, double* sH, double* sV, unsigned* iA, void* tmp, size_t tmp_bytes, double* knots, int* M,
                               double* ab, double* dth, int accumulate);
size_t interp_sort_temp_bytes(long long nd_max);
int launch_interp_theta(hipStream_t st, const LawDev& L, double T, int n_half, const double* nodeH, const double* nodeV,
                        long long nd, double* sH, double* sV, void* tmp, size_t tmp_bytes, double* knots, int* M, double* G,
                        double* ab, double* dth, int accumulate);

// U law (SIA2D_D_target(interpolation = :Linear), target_D_pure.jl:179-193): bilinear on the fixed node grid of Laws.jl:128-169
int launch_interp_theta_U(hipStream_t st, const LawDev& L, int n_half, const double* nodeH, const double* nodeS, const double* nodeV,
                          long long nd, double* sA, double* sB, void* tmp, size_t tmp_bytes, double* cell4, double* G, int* err,
                          double* dth, int accumulate);

// k_vel.hip (A-type law modes 0 / 1; lm >= 2: the U law of target :D)
struct VArgs;
void launch_surface_V(int lm, int nblk, hipStream_t st, Pools P, LawDev L, const double* U, double* Vx, double* Vy, int base,
                      double finv);
void launch_surfV_vjp(int lm, int mode, int nblk, hipStream_t st, Pools P, LawDev L, const VArgs& A, int base);
void launch_avgv_axpy(int nblk, hipStream_t st, Pools P, const double* Vx, const double* Vy, double* ax, double* ay, const double* w);
void launch_avgv_cot(int nblk, hipStream_t st, Pools P, double* ax, double* ay, const double* Vabs, const double* Vxr,
                     const double* Vyr, const unsigned char* on, int component_abs, double weight);

// k_misc.hip
void launch_controller(int G, hipStream_t st, Pools P, CtrlArgs C);
void launch_poststep(int nblk, hipStream_t st, Pools P, PostArgs A, double* Ua, double* Ub);
void launch_sum_part(int ng, hipStream_t st, Pools P, int slot, double* out, int accumulate, int g0);
void launch_sum_part_steps(int ng, hipStream_t st, Pools P, const double* base, long long stride, int jhi, int jlo,
                           int slot, double* out);
void launch_sum_part_theta(int Pn, int ng, hipStream_t st, Pools P, const double* part_theta, double* out,
                           int accumulate, int g0);
void launch_loss(int nblk, hipStream_t st, Pools P, const double* H, const double* Href, const unsigned char* mask,
                 const double* ws, const int* refslot, long long ntot, double log_eps);
void launch_mb_vjp(int nblk, hipStream_t st, Pools P, const double* Hpre, const double* mb0, const double* Sref,
                   const double* lam_in, double* lam_out, int add, int base, const int* gflag = nullptr,
                   const int* gslot = nullptr, long long ntot = 0);
void launch_mb_apply(int nblk, hipStream_t st, Pools P, const double* H, const double* mb0, const double* Sref,
                     double* Hn, double* MBout, int base);
void launch_dhdt_sums(int nblk, int G, hipStream_t st, Pools P, const double* snaps, const int* i0s, const int* i1s, long long ntot,
                      double* part2, const double* dts, const double* refs, double w, double* coef, double* lossacc);
void launch_dhdt_cot(int nblk, hipStream_t st, Pools P, double* lam, const double* snaps, const int* i0s, const int* i1s,
                     const double* coef, int j, long long ntot);
void launch_law_field(hipStream_t st, LawDev L, const double* T, double* Aout, long long n);
int launch_law_field_grad(hipStream_t st, LawDev L, const double* T, const double* G, long long n, double* part_theta, int max_rows);
void launch_law_field_grad_scratch(int nblk, hipStream_t st, LawDev L, const double* T, const double* G, long long n,
                                   double* gscratch, double* part_theta);
void launch_sum_rows(int Pn, hipStream_t st, const double* part, int nrows, double* out);
void launch_sum_slots(hipStream_t st, long long n, int nslots, const double* slots, double* out);
void launch_utab_build(hipStream_t st, LawDev L, double* tab, int nh, int ns, double floor_abs, unsigned long long* stat);
void launch_ytab_build(hipStream_t st, Pools P, LawDev L, int G, double* tab, int ni, double floor_abs, unsigned long long* stat);
void launch_eval_law(hipStream_t st, Pools P, LawDev L, const double* U, double* out, int gidx, long long nd);
void launch_axpy_g(int nblk, hipStream_t st, Pools P, const double* x, const double* y, double* z);
void launch_axpy(long long n, hipStream_t st, double a, const double* x, const double* y, double* z);  // z = y + a x
void launch_lerp(long long n, hipStream_t st, double s, const double* a, const double* b, double* out);  // out = a + s (b - a)
void launch_lerp_g(int nblk, hipStream_t st, Pools P, const double* snaps, long long ntot, const int* seg, const double* sw, double* out);
void launch_seg_pairs(long long ntot, int n_seg, hipStream_t st, const double* snaps, double2* segs);
void launch_sum_tilesFt(int G, int rows, hipStream_t st, Pools P, const double* part, double* out);
// VelocityRegularization (Regularization.jl:192-245): see k_vreg_* in sia2d_device.hpp
void launch_vreg_prep(int nblk, hipStream_t st, Pools P, const double* H, const double* vx, const double* vy, const double* w,
                      int dist, double* Vabs, unsigned char* mask);
void launch_vreg_lap(int nblk, hipStream_t st, Pools P, const double* Vabs, const unsigned char* mask, const double* w, double* r);
void launch_vreg_cot(int nblk, hipStream_t st, Pools P, const double* r, const double* Vabs, const double* w, double* vx, double* vy);
void launch_initdt_norms(int nblk, hipStream_t st, Pools P, const double* U, const double* F0, const double* F1,
                         double abstol, double reltol);
void launch_initdt_ctrl(int G, hipStream_t st, Pools P, int phase, double tspan, double dtmax, double* dt0store);
void launch_begin(int G, hipStream_t st, Pools P, const double* tstops, double dtmax, double dt_given);
void launch_set_dt(int G, hipStream_t st, Pools P, double dt);
void launch_vref_itp(int nblk, hipStream_t st, Pools P, VItpArgs A);
void launch_vq_finish(int G, hipStream_t st, Pools P, const AdjState* adj, const int* slotA, int scale_loss, double wq, double* out,
                      double* coef);
void launch_surfV_theta_node(int lm, int nblk, hipStream_t st, Pools P, VItpArgs I, const double* snaps, int component_abs, double log_eps,
                             double* tnode);
void launch_surfV_theta_only(int lm, int nblk, hipStream_t st, Pools P, const VArgs& A);
void launch_gacc_axpy(int nblk, hipStream_t st, Pools P, const double* coef, const double* tnode, double* Gacc);
void launch_vref_scale(int G, hipStream_t st, Pools P, const AdjState* adj, const int* slotA, int scale_loss, double wq,
                       double* scale_out, double* w_out);
void launch_adj_begin(int G, hipStream_t st, Pools P, AdjState* adj, const int* n_snaps, double tau0, const int* mb_flags,
                      const int* mb_slots);
void launch_adj_itp(int G, hipStream_t st, Pools P, AdjState* adj, const double* tsnap, const int* n_snaps, int all_at_end);
void launch_tikhonov(hipStream_t st, const double* a, const unsigned char* mask, double* r, double* grad,
                     double* partial, int nx, int ny, double dx, double dy);
void launch_adj_poststep(int nblk, hipStream_t st, Pools P, AdjPostArgs A, double* Ua, double* Ub);

}  // namespace odinn
