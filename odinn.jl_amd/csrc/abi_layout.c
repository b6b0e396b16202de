/* abi_layout.c -- prints size and field offsets of every struct of include/odinn_hip.h as the C compiler lays them out:
 *   "S <struct> <sizeof>" and "F <struct> <field> <offsetof> <sizeof field>" lines.
 * Built by `make abi_layout` (gcc, host only; __graft_entry__.build() runs it); tests/test_abi.py compares the table with the
 * ctypes mirrors of odinn.jl_amd/_lib.py and with the struct mirrors of julia/OdinnHIP.jl, field by field.
 * The _Static_asserts pin the layout the committed bindings were written against: changing a struct breaks the build here first. */
#include <stddef.h>
#include <stdio.h>

#include "../../include/odinn_hip.h"

#define S(T) printf("S %s %zu\n", #T, sizeof(T))
#define F(T, f) printf("F %s %s %zu %zu\n", #T, #f, offsetof(T, f), sizeof(((T*)0)->f))

_Static_assert(sizeof(odinn_phys) == 72, "odinn_phys");
_Static_assert(sizeof(odinn_glacier_desc) == 112 && offsetof(odinn_glacier_desc, phys) == 24, "odinn_glacier_desc");
_Static_assert(sizeof(odinn_mlp_desc) == 136 && offsetof(odinn_mlp_desc, pre_lo) == 80 && offsetof(odinn_mlp_desc, post_lo) == 120,
               "odinn_mlp_desc");
_Static_assert(sizeof(odinn_solver_opts) == 64 && offsetof(odinn_solver_opts, cfl) == 56, "odinn_solver_opts");
_Static_assert(sizeof(odinn_solve_stats) == 40, "odinn_solve_stats");
_Static_assert(sizeof(odinn_adjoint_opts) == 40 && offsetof(odinn_adjoint_opts, maxiters) == 32, "odinn_adjoint_opts");
_Static_assert(sizeof(odinn_schedule) == 80 && offsetof(odinn_schedule, adj_ut_fused) == 72 && offsetof(odinn_schedule, reserved) == 76, "odinn_schedule");

int main(void) {
  S(odinn_phys);
  F(odinn_phys, rho); F(odinn_phys, g); F(odinn_phys, eta0); F(odinn_phys, n); F(odinn_phys, p); F(odinn_phys, q);
  F(odinn_phys, C); F(odinn_phys, minA); F(odinn_phys, maxA);
  S(odinn_glacier_desc);
  F(odinn_glacier_desc, nx); F(odinn_glacier_desc, ny); F(odinn_glacier_desc, dx); F(odinn_glacier_desc, dy);
  F(odinn_glacier_desc, phys); F(odinn_glacier_desc, A); F(odinn_glacier_desc, T);
  S(odinn_mlp_desc);
  F(odinn_mlp_desc, n_layers); F(odinn_mlp_desc, widths); F(odinn_mlp_desc, acts); F(odinn_mlp_desc, has_prescale);
  F(odinn_mlp_desc, pre_lo); F(odinn_mlp_desc, pre_hi); F(odinn_mlp_desc, post_kind); F(odinn_mlp_desc, post_lo);
  F(odinn_mlp_desc, post_hi);
  S(odinn_solver_opts);
  F(odinn_solver_opts, reltol); F(odinn_solver_opts, abstol); F(odinn_solver_opts, dtmax); F(odinn_solver_opts, dt0);
  F(odinn_solver_opts, fixed_dt); F(odinn_solver_opts, maxiters); F(odinn_solver_opts, scheme); F(odinn_solver_opts, dense);
  F(odinn_solver_opts, cfl);
  S(odinn_solve_stats);
  F(odinn_solve_stats, naccept); F(odinn_solve_stats, nreject); F(odinn_solve_stats, nrhs); F(odinn_solve_stats, t_final);
  F(odinn_solve_stats, dt_last);
  S(odinn_adjoint_opts);
  F(odinn_adjoint_opts, reltol); F(odinn_adjoint_opts, abstol); F(odinn_adjoint_opts, dtmax); F(odinn_adjoint_opts, n_quadrature);
  F(odinn_adjoint_opts, reserved); F(odinn_adjoint_opts, maxiters);
  S(odinn_schedule);
  F(odinn_schedule, step_sc); F(odinn_schedule, fused_tiles); F(odinn_schedule, dhdt_strip); F(odinn_schedule, vjph_strip);
  F(odinn_schedule, vjpth_strip); F(odinn_schedule, snap_on_load); F(odinn_schedule, interp_streams);
  F(odinn_schedule, interp_batch); F(odinn_schedule, lawgrad_wave); F(odinn_schedule, vq_onepass); F(odinn_schedule, adj_fused);
  F(odinn_schedule, adj_skip); F(odinn_schedule, adj_segs); F(odinn_schedule, adj_rows); F(odinn_schedule, adj_theta_fused);
  F(odinn_schedule, law_table); F(odinn_schedule, interp_async); F(odinn_schedule, adj_sc); F(odinn_schedule, adj_ut_fused); F(odinn_schedule, reserved);
  return 0;
}
