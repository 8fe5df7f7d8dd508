/*
 * rays_hip.h -- C ABI of the MI355X-native RAYS ray-trajectory integrator (librays_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of ORNL-Fusion/RAYS:
 *
 *     call trace_rays            RAYS_project/RAYS_code/RAYS.f90:13
 *                                RAYS_project/RAYS_lib/ray_tracing.f90:1-290
 *
 * i.e. the per-ray ODE march  ode_solver(RK4_ode | SG_ode) -> eqn_ray ->
 * equilibrium(slab | solovev) + deriv_cold | deriv_num -> check_save, for all rays of a fan.
 * The reference has no process/device boundary (module state in, ray_results_m arrays out);
 * the binding a RAYS maintainer adds is the iso_c_binding interface in fortran/rays_hip_m.f90
 * plus the replacement trace_rays in fortran/trace_rays_hip.f90.
 *
 * Conventions
 *   - plain C types only; every array is caller-owned; the library never keeps a host pointer
 *     after a call returns.
 *   - return 0 on success; non-zero = HIP failure / bad configuration, message via
 *     rays_hip_last_error().  A ray that stops is NOT an error: it gets a stop code
 *     (RAYS_STOP_*), mapped to the reference's ode_stop_flag text by rays_hip_stop_flag_text().
 *   - all reals are IEEE binary64 (reference rkind = selected_real_kind(15,307),
 *     constants_m.f90:14), counters are 32-bit.
 *   - array layouts are exactly the reference's Fortran arrays (ray_results_m.f90:44-58):
 *       ray_vec (nv, nstep_max+1, nray)  column-major == C [nray][nstep_max+1][nv]
 *       residual(nstep_max+1, nray)      column-major == C [nray][nstep_max+1]
 *       rvec0(3,nray), rindex_vec0(3,nray)             == C [nray][3]   (ray_init_m.f90:47-53)
 *     Entries past npoints(iray) are zero, as after initialize_ray_results_m (154-164).
 */
#ifndef RAYS_HIP_H
#define RAYS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAYS_ABI_VERSION 3
#define RAYS_NSPEC0 5 /* species_m.f90:25  nspec0; arrays are dimensioned 0:nspec0 */
#define RAYS_NS0 (RAYS_NSPEC0 + 1)

/* ---- selectors: integer images of the reference's string switches ----------------------- */
enum { RAYS_ODE_RK4 = 0, RAYS_ODE_SG = 1 };           /* ode_m.f90:238  'RK4_ODE' | 'SG_ODE'   */
enum { RAYS_DERIV_COLD = 0, RAYS_DERIV_NUM = 1 };     /* eqn_ray.f90:106 'cold' | 'numerical'  */
enum { RAYS_PARAM_ARCL = 0, RAYS_PARAM_TIME = 1 };    /* eqn_ray.f90:148 'arcl' | 'time'       */
enum { RAYS_EQ_SLAB = 0, RAYS_EQ_SOLOVEV = 1, RAYS_EQ_AXISYM = 2 }; /* equilibrium_m.f90:177-189 */
enum { RAYS_DAMP_NONE = 0, RAYS_DAMP_FUND_ECH = 1 };  /* damping_m.f90:94-101 'no_damp' | 'damp_fund_ECH' */

/* slab_eq_m.f90:172-300 profile-model strings */
enum { RAYS_SLAB_BX_ZERO = 0 };
enum { RAYS_SLAB_BY_ZERO = 0, RAYS_SLAB_BY_CONSTANT, RAYS_SLAB_BY_TOROID, RAYS_SLAB_BY_LINEAR_SHEAR };
enum { RAYS_SLAB_BZ_CONSTANT = 0, RAYS_SLAB_BZ_TOROID, RAYS_SLAB_BZ_LINEAR, RAYS_SLAB_BZ_LINEAR_2 };
enum { RAYS_SLAB_N_CONSTANT = 0, RAYS_SLAB_N_LINEAR, RAYS_SLAB_N_LINEAR_2, RAYS_SLAB_N_PARABOLIC,
       RAYS_SLAB_N_GAUSSIAN };
enum { RAYS_SLAB_T_ZERO = 0, RAYS_SLAB_T_CONSTANT, RAYS_SLAB_T_LINEAR, RAYS_SLAB_T_LINEAR_2,
       RAYS_SLAB_T_PARABOLIC };
/* solovev_eq_m.f90:208-267 */
enum { RAYS_SOLOVEV_N_CONSTANT = 0, RAYS_SOLOVEV_N_PARABOLIC };
enum { RAYS_SOLOVEV_T_ZERO = 0, RAYS_SOLOVEV_T_PARABOLIC = 2 };

/* axisym_toroid_eq_m.f90:56-100: magnetics / profile model strings */
enum { RAYS_AXI_MAG_EQDSK_SPLINE = 0, /* 'eqdsk_magnetics_spline_interp' */
       RAYS_AXI_MAG_SOLOVEV = 1 };    /* 'solovev_magnetics' (solovev_magnetics_m.f90): the analytic Solovev field under
                                         the axisym_toroid profile models; its /solovev_magnetics_list/ travels in
                                         rays_params_t.solovev (rmaj, kappa, bphi0, iota0, outer_bound = outer_boundary,
                                         psiB, box_*), and axisym.box_* / psiB repeat the box and psiB */
enum { RAYS_AXI_MAG_EQDSK_LIN = 2 };  /* 'eqdsk_magnetics_lin_interp' (eqdsk_magnetics_lin_interp_m.f90): bilinear
                                         interpolation of the raw eqdsk Psi(NRBOX, NZBOX) and linear interpolation of
                                         T(NRBOX), derivatives by the reference's central differences over +-dR, +-dZ
                                         (eqdsk_utilities_m.f90:144-306); tables: rays_hip_set_eqdsk_lin_tables */
enum { RAYS_AXI_N_CONSTANT = 0, RAYS_AXI_N_PARABOLIC, RAYS_AXI_N_SPLINE };
enum { RAYS_AXI_T_ZERO = 0, RAYS_AXI_T_CONSTANT, RAYS_AXI_T_PARABOLIC, RAYS_AXI_T_SPLINE };

/* ---- per-ray stop codes  <->  reference ode_stop_flag strings ---------------------------- */
enum {
  RAYS_STOP_NONE = 0,
  RAYS_STOP_SOUT_GT_SMAX = 1,      /* 'sout > s_max'            ray_tracing.f90:144 */
  RAYS_STOP_NSTEP_MAX = 2,         /* ' nstep > nstep_max'      ray_tracing.f90:152 (leading blank) */
  RAYS_STOP_X_OUT_OF_BOUNDS = 10,  /* 'x out_of_bounds'         slab_eq_m.f90:163 */
  RAYS_STOP_Y_OUT_OF_BOUNDS = 11,  /* 'y out_of_bounds'         slab_eq_m.f90:164 */
  RAYS_STOP_Z_OUT_OF_BOUNDS = 12,  /* 'z out_of_bounds'         slab_eq_m.f90:165 */
  RAYS_STOP_NEGATIVE_DENS = 13,    /* 'negative_dens'           slab_eq_m.f90:305, solovev_eq_m.f90:272 */
  RAYS_STOP_NEGATIVE_TEMP = 14,    /* 'negative_temp'           slab_eq_m.f90:306, solovev_eq_m.f90:273 */
  RAYS_STOP_R_OUT_OF_BOX = 20,     /* 'R out_of_box'            solovev_eq_m.f90:155 */
  RAYS_STOP_Z_OUT_OF_BOX = 21,     /* 'z out_of_box'            solovev_eq_m.f90:156 */
  RAYS_STOP_AXI_R_OUT_OF_BOX = 22, /* 'R_out_of_box'            axisym_toroid_eq_m.f90:263 */
  RAYS_STOP_AXI_Z_OUT_OF_BOX = 23, /* 'Z_out_of_box'            axisym_toroid_eq_m.f90:267 */
  RAYS_STOP_OUT_OF_PLASMA = 24,    /* 'out_of_plasma'           axisym_toroid_eq_m.f90:288 */
  RAYS_STOP_SOLMAG_R_OUT_OF_BOUNDS = 25, /* 'R out_of_bounds'   solovev_magnetics_m.f90:147 */
  RAYS_STOP_SOLMAG_Z_OUT_OF_BOUNDS = 26, /* 'z out_of_bounds'   solovev_magnetics_m.f90:148 */
  RAYS_STOP_INFINITE_VG_RHS = 30,  /* 'infinite Vg'             eqn_ray.f90:142 */
  RAYS_STOP_RAY_STALLED = 31,      /* 'ray stalled'             eqn_ray.f90:168 */
  RAYS_STOP_DISP_RESIDUAL = 40,    /* 'dispersion_residual'     check_save.f90:70 */
  RAYS_STOP_INFINITE_VG_CHECK = 41,/* 'infinite_Vg'             check_save.f90:108 */
  RAYS_STOP_TOTAL_ABSORPTION = 42, /* 'total_absorption'        check_save.f90:123 */
  RAYS_STOP_ODE_TOTAL_ERROR = 50,  /* 'ODE total error'         SG_ode_m.f90:144 */
  RAYS_STOP_SG_MAXNUM = 51,        /* 'step number .ge. maxnum' ode_RAYS.f90:538 */
  RAYS_STOP_SG_STIFF = 52,         /* 'equations stiff'         ode_RAYS.f90:541 */
  RAYS_STOP_SG_T_EQ_TOUT = 53,     /* 't == tout'               ode_RAYS.f90:433 */
  RAYS_STOP_SG_NEG_ERR = 54,       /* 'relerr or abserr < 0'    ode_RAYS.f90:439 */
  RAYS_STOP_SG_EPS_LE_0 = 55       /* 'eps <= 0'                ode_RAYS.f90:447 */
};

/* ---- slab_eq_m.f90:35-85 namelist /slab_eq_list/ ------------------------------------------ */
typedef struct rays_slab_params {
  int32_t bx_prof_model, by_prof_model, bz_prof_model, dens_prof_model;
  int32_t t_prof_model[RAYS_NS0];
  int32_t pad_[2];
  double xmin, xmax, ymin, ymax, zmin, zmax;
  double rmaj, rmin, x0;
  double bx0, by0, bz0, LBy_shear_scale, LBz_scale, dBzdx;
  double Ln_scale, dndx, alphan1, alphan2, n_min;
  double LT_scale, dtdx;
  double alphat1[RAYS_NS0], alphat2[RAYS_NS0], T_min[RAYS_NS0];
} rays_slab_params_t;

/* ---- solovev_eq_m.f90:17-45 namelist /solovev_eq_list/ + derived psiB (:92) --------------- */
typedef struct rays_solovev_params {
  int32_t dens_prof_model;
  int32_t t_prof_model[RAYS_NS0];
  int32_t pad_[1];
  double rmaj, kappa, bphi0, iota0, outer_bound;
  double psiB; /* computed by the host exactly as solovev_eq_m.f90:89-92 */
  double alphan1, alphan2;
  double alphat1[RAYS_NS0], alphat2[RAYS_NS0];
  double box_rmin, box_rmax, box_zmin, box_zmax;
} rays_solovev_params_t;

/* ---- axisym_toroid_eq_m.f90:56-100 namelist /axisym_toroid_eq_list/ + eqdsk-derived scalars ---- */
typedef struct rays_axisym_params {
  int32_t magnetics_model, density_prof_model;
  int32_t t_prof_model[RAYS_NS0];
  double box_rmin, box_rmax, box_zmin, box_zmax; /* from the eqdsk (eqdsk_magnetics_spline_interp_m.f90:115-118) */
  double plasma_psi_limit;
  double psiB; /* PSIBOUND - PSIAXIS (eqdsk_magnetics_spline_interp_m.f90:171-173) */
  double alphan1, alphan2, d_scrape_off, T_scrape_off;
  double alphat1[RAYS_NS0], alphat2[RAYS_NS0];
} rays_axisym_params_t;

/* ---- everything trace_rays reads from module state ----------------------- */
typedef struct rays_params {
  int32_t abi_version;  /* RAYS_ABI_VERSION */
  int32_t nv;           /* ode_m.f90:160-173: 7, +1 with damping, +1+nspec with multi_spec_damping, +5 with
                           integrate_eq_gradients */
  int32_t nspec;        /* species_m.f90:27 number of ion species (electrons are species 0) */
  int32_t nstep_max;    /* ode_m.f90:104 */
  int32_t ode_solver;   /* RAYS_ODE_* */
  int32_t ray_deriv;    /* RAYS_DERIV_* */
  int32_t ray_param;    /* RAYS_PARAM_* */
  int32_t equilib_model;/* RAYS_EQ_* */
  int32_t integrate_eq_gradients; /* diagnostics_m.f90:101 */
  int32_t pad_[3];
  double ds, s_max;                    /* ode_m.f90:98-101 */
  double omgrf, k0;                    /* rf_m.f90:18-21 (passed by value: never re-derived) */
  double clight, eps0;                 /* constants_m.f90:42-44 (single-precision literals!) */
  double dispersion_resid_limit;       /* rf_m.f90:48 */
  double rel_err0, abs_err0, SG_error_limit; /* SG_ode_m.f90:26-31 */
  double qs[RAYS_NS0], ms[RAYS_NS0];   /* species_m.f90:71-72, SI units after init (155-157) */
  double n0s[RAYS_NS0], t0s[RAYS_NS0]; /* species_m.f90:42-44 */
  double eta[RAYS_NS0];                /* species_m.f90:73 */
  rays_slab_params_t slab;
  rays_solovev_params_t solovev;
  /* damping_m.f90:30-40 (appended in ABI version 2) */
  int32_t damping_model;      /* RAYS_DAMP_* */
  int32_t multi_spec_damping; /* damping_m.f90:35: one absorbed-power row per species behind the total */
  double total_damping_limit; /* damping_m.f90:38 */
  /* appended in ABI version 3 */
  rays_axisym_params_t axisym;
} rays_params_t;

/* ---- library control ----------------------------------------------------------------------- */
/* Select up to ngpu visible devices for rays_hip_trace (0 = all).  Returns the number in use,
 * or a negative value on failure.  Optional: rays_hip_trace initialises lazily. */
int rays_hip_init(int ngpu);
/* Explicit device list for rays_hip_trace: slot i of the ray partition runs on device_ids[i]
 * (1 <= n <= 16).  A device may be listed more than once: its slots then run concurrently on that
 * device, each with its own host thread, stream and staging buffers, so one slot's device-to-host
 * copy overlaps another's trace.  Returns n, or a negative value on failure. */
int rays_hip_init_devices(int n, const int* device_ids);
int rays_hip_finalize(void);
int rays_hip_device_count(void);
/* sizeof(rays_params_t) as compiled into the library: lets a foreign-language binding (ctypes,
 * iso_c_binding) verify its mirror of the struct. */
int rays_hip_sizeof_params(void);
/* Copies the last error message (NUL-terminated, truncated to len) and returns its length. */
int rays_hip_last_error(char* buf, int len);
/* Reference ode_stop_flag text for a stop code (e.g. " nstep > nstep_max"); "" if unknown. */
const char* rays_hip_stop_flag_text(int stop_code);
/* Z-function spline table for damping_model = 'damp_fund_ECH' (zfunctions_m.f90:436-466): the host
 * owns the table (the reference builds it in initialize_spline_coeffs); fspl_re is the PPPL-pspline
 * compact cubic spline fsplRe(4, nx) in Fortran order == C [nx][4], on the uniform grid
 * x_min .. x_max.  The library copies it; it must be set before tracing with damping. */
int rays_hip_set_zfun_table(const double* fspl_re, int nx, double x_min, double x_max);

/* Spline tables of equilib_model = 'axisym_toroid' with magnetics_model =
 * 'eqdsk_magnetics_spline_interp'.  Coefficient generation (PPPL pspline bcspline/cspline, not-a-knot)
 * stays on the host -- in RAYS it is initialize_eqdsk_magnetics_spline_interp /
 * initialize_density_spline_interp / initialize_temperature_spline_interp -- and only evaluation
 * (bcspeval / cspeval on uniform grids) runs on the device.  All arrays are the host objects'
 * own storage (type cube_spline_function_1D/2D, quick_cube_splines_m.f90:36-56):
 *   psi_fspl(4,4,nr,nz) Fortran order, on r_grid(nr) x z_grid(nz);   Psi - PSIAXIS
 *   rb_fspl(4,n_rb) on rb_grid:   T = R*Bphi
 *   ne/te/ti_fspl(4,n) on *_grid (psiN 0..1): profiles normalised to 1 on axis; n = 0 if unused.
 * magnetics_model = 'solovev_magnetics' (analytic field) has no psi / R*Bphi tables: the call is needed only
 * when n or T are splined, with nr = nz = n_rb = 0 and the profile tables alone.
 * The library copies everything; set before tracing. */
typedef struct rays_axisym_tables {
  int32_t nr, nz, n_rb, n_ne, n_te, n_ti;
  const double *r_grid, *z_grid, *psi_fspl;
  const double *rb_grid, *rb_fspl;
  const double *ne_grid, *ne_fspl;
  const double *te_grid, *te_fspl;
  const double *ti_grid, *ti_fspl;
} rays_axisym_tables_t;
int rays_hip_set_axisym_tables(const rays_axisym_tables_t* t);

/* Tables of magnetics_model = 'eqdsk_magnetics_lin_interp' (instead of rays_hip_set_axisym_tables): the host's
 * eqdsk_utilities_m arrays after initialize_eqdsk_magnetics_lin_interp (eqdsk_magnetics_lin_interp_m.f90:121-141),
 * in the same struct with this meaning:
 *   r_grid(nr) = R_grid, z_grid(nz) = Z_grid;   psi_fspl = Psi(nr, nz) Fortran order, PSIAXIS already subtracted;
 *   rb_fspl = T(nr) (= R*Bphi on R_grid), n_rb = nr, rb_grid unused (may be NULL);
 *   ne/te/ti_*: the splined profiles as in rays_hip_set_axisym_tables (n = 0 if unused);
 *   dR, dZ = eqdsk_utilities_m's dR, dZ (HALF the grid spacing, :137-138) -- the steps of GetPsiR .. GetRBphiR.
 * axisym.box_* and axisym.psiB (= PSIBOUND - PSIAXIS) travel in rays_params_t as for the spline model.
 * Where the reference's central differences index one cell beyond R_grid / Z_grid (a point in the outermost cells)
 * it reads the neighbouring column of Psi through Fortran's storage order; the library does the same and clamps only
 * what would leave the array altogether (undefined in the reference). */
int rays_hip_set_eqdsk_lin_tables(const rays_axisym_tables_t* t, double dR, double dZ);

/* Numerics of the trace kernels (process-wide; read when a trace is launched).
 *   RAYS_NUMERICS_EXACT (default): IEEE binary64 in the reference's operation order, no FMA contraction,
 *     correctly rounded quotients and roots -- trajectories bit-identical to the reference CPU path.
 *   RAYS_NUMERICS_TOLERANCE: NOT bit-identical -- the kernels fuse a*b+c, re-associate, use once-refined reciprocals
 *     and roots and algebraic short cuts.  What it is was measured on the full BASELINE fans against the reference
 *     (tests/test_gpu_numerics_full_fans.py, profiles/numerics_evidence.json): restarted from every recorded reference
 *     point, every step lands within 1e-10 relative (norm-wise on r and k) of the reference's next point -- the bar of
 *     BASELINE.json's north_star; measured 4.4e-15 over the headline fan's 12 873 661 steps, 1e-12 over 32.8 M steps of
 *     the slab fan -- and ray counts / step indices / stop flags are exactly the reference's on every ray surveyed;
 *     ACCUMULATED along a ray the trajectory deviates by up to 1.6e-8.  The steps where that bar cannot hold for any
 *     arithmetic but the reference's own (a ray's last recorded step before two modes coalesce) are handed over to a
 *     kernel of the exact build.  Built for ode_solver = RK4 with ray_deriv = cold (without
 *     multi_spec_damping); every other configuration runs its exact kernel under either setting (finite-difference
 *     dD amplifies an ulp by 1e8, and the adaptive solver then takes another step sequence).  Fans of 131072 rays and
 *     more run a two-waves-per-SIMD build without the hand-over: on Solovev fans that large a ray's last recorded step
 *     can deviate by up to 4.5e-10.
 * The environment variable RAYS_HIP_NUMERICS = exact | tolerance sets the initial value.  Returns the previous
 * setting, or -1 for an unknown mode. */
enum { RAYS_NUMERICS_EXACT = 0, RAYS_NUMERICS_TOLERANCE = 1 };
int rays_hip_set_numerics(int mode);
int rays_hip_get_numerics(void);

/* Validates a parameter block exactly as the reference's `stop 1` configuration checks would;
 * 0 if the device path supports it. */
int rays_hip_check_params(const rays_params_t* p);

/* ---- the hot path, host-pointer form: replaces `call trace_rays` ---------------------------
 * Blocking.  Shards rays in contiguous blocks over the selected GPUs (the reference's
 * OpenMP `schedule(static)`, ray_tracing.f90:62), copies each device's contiguous output slab
 * straight into the caller's arrays.  Optional outputs may be NULL.
 * ray_vec / residual: points 1..npoints(i) of every ray are written; entries beyond are left as the
 * caller passed them -- as in the reference, whose arrays are zero-filled by
 * initialize_ray_results_m (ray_results_m.f90:154-164) before trace_rays runs.  (Only the
 * recorded points cross PCIe: packed on the device, unpacked by host threads.)  One caller thread at a
 * time, like the reference's trace_rays; the entry keeps its device buffers, pinned staging and stream
 * between calls until rays_hip_finalize().
 *   stop_code[nray]          integer image of ray_stop_flag(iray)          (ray_tracing.f90:258)
 *   end_ray_vec[nray][nv]    v at the last valid step                       (ray_tracing.f90:260)
 *   end_residuals[nray]      residual(nstep,iray)                           (ray_tracing.f90:255)
 *   max_residuals[nray]      maxval(abs(residual(1:nstep,iray)))            (ray_tracing.f90:256)
 *   elapsed_s                wall seconds of the device work incl. copies
 */
int rays_hip_trace(const rays_params_t* p, int nray,
                   const double* rvec0, const double* rindex_vec0,
                   double* ray_vec, double* residual, int32_t* npoints, int32_t* stop_code,
                   double* end_ray_vec, double* end_residuals, double* max_residuals,
                   double* elapsed_s);

/* ---- the hot path, device-pointer form ------------------------------------------------------
 * All pointers are device memory on the CURRENT HIP device; the launch is asynchronous on
 * `hip_stream` (a hipStream_t, NULL = default stream).  ray_vec/residual are zero-filled by the
 * call (hipMemsetAsync on the same stream) unless RAYS_TRACE_NO_ZERO_FILL is set in flags.
 * Used by the Python host (torch tensors), by bench.py, and by rays_hip_trace itself.
 */
enum { RAYS_TRACE_NO_ZERO_FILL = 1 };
int rays_hip_trace_device(const rays_params_t* p, int nray,
                          const double* d_rvec0, const double* d_rindex_vec0,
                          double* d_ray_vec, double* d_residual, int32_t* d_npoints,
                          int32_t* d_stop_code, double* d_end_ray_vec, double* d_end_residuals,
                          double* d_max_residuals, void* hip_stream, int flags);

/* ---- ray_scan fused into ONE launch ---------------------------------------
 * Replaces the reference's serial scan loop (ray_scan/ray_scan.f90:33-49: update_scan_parameter +
 * initialize + trace_rays per run; scanner_m.f90:174-205: scan_parameter = 'ds', the only physical one)
 * by one launch over n_runs x nray rays: run r traces the fan rvec0/rindex_vec0[nray][3] with
 * ds = d_ds_values[r]; everything else comes from p.  Outputs are the arrays of rays_hip_trace_device
 * with a leading run dimension: ray_vec[n_runs][nray][nstep_max+1][nv], npoints[n_runs][nray], ...
 * Each run's results are those of a stand-alone trace with that ds, bit for bit. */
int rays_hip_scan_device(const rays_params_t* p, int n_runs, const double* d_ds_values, int nray,
                         const double* d_rvec0, const double* d_rindex_vec0, double* d_ray_vec,
                         double* d_residual, int32_t* d_npoints, int32_t* d_stop_code,
                         double* d_end_ray_vec, double* d_end_residuals, double* d_max_residuals,
                         void* hip_stream, int flags);

/* ---- one output step from arbitrary states: the ode_m interface --------------------------------
 * Batched image of `call ode_solver(eqn_ray, nv, v, s, sout, ray_stop)` (ode_m.f90:218-254, with
 * sout = s + ds and, for SG_ODE, ray_stop%rel_err/abs_err at rel_err0/abs_err0 as after
 * ray_init_ode_solver, ode_m.f90:182-214) followed by the check_save trace_rays applies to the new point
 * (ray_tracing.f90:212-243).  d_v0[n][nv] states, d_s0[n] ray parameters (NULL = 0) -> d_v1[n][nv],
 * d_resid[n] (may be NULL), d_stop_code[n]: RAYS_STOP_NONE when the step was taken and kept, then v1 is the new
 * state.  Otherwise v1 is what ode_solver left in v: the advanced state when check_save refused the step
 * (ray_tracing.f90:214-234), v0 when the solver itself stopped (RK4_ode_m.f90:83-89), zeros when v0 already failed
 * the initial check_save (such a ray never starts: ray_tracing.f90:100-112) -- a recorded trajectory point always
 * passes it.  ASYNCHRONOUS on hip_stream like rays_hip_trace_device (it was blocking before round 3): synchronise the
 * stream -- hipStreamSynchronize, or hipDeviceSynchronize for the null stream -- before reading d_v1 / d_resid /
 * d_stop_code on the host or from another stream.  Its scratch (one block per device and stream) is kept between
 * calls and released by rays_hip_finalize; ONE caller thread per (device, stream): a second thread that asks the same
 * stream for a larger block frees the first one's behind hipStreamSynchronize, which does not cover work the first
 * thread has not enqueued yet. */
int rays_hip_ode_step_device(const rays_params_t* p, int n, const double* d_v0, const double* d_s0,
                             double* d_v1, double* d_resid, int32_t* d_stop_code, void* hip_stream);

/* Name of the kernel specialisation rays_hip_trace_device would launch for p (for profiling
 * scripts: matches the rocprofv3 kernel-trace name prefix). */
const char* rays_hip_kernel_name(const rays_params_t* p);
/* Same for a fan of nray rays: large fans may get a differently tuned build of the kernel. */
const char* rays_hip_kernel_name_for(const rays_params_t* p, int nray);

/* ---- multi-GPU trace with a device-resident result: the final trajectory gather over RCCL -------
 * (north_star: "RCCL over xGMI used only for the final trajectory gather".)
 * Shards the rays in contiguous blocks over the devices chosen by rays_hip_init[_devices] (distinct devices;
 * one host thread per device, like rays_hip_trace), traces every block on its device, then gathers the blocks'
 * PACKED trajectories and per-ray summaries on the root (= the first device of the list) with one grouped batch of
 * ncclSend/ncclRecv -- every peer uses its own xGMI link to the root -- and unpacks them there into the padded
 * reference layout.  The result stays in device memory on the root, owned by the library and valid until the
 * next rays_hip_trace_gather call or rays_hip_finalize; follow-on device work (rays_hip_deposition_device) can
 * consume it in place, rays_hip_result_to_host copies it out.  librccl.so is loaded on the first call with more
 * than one device.  rvec0 / rindex_vec0 are host arrays ([nray][3]). */
typedef struct rays_device_result {
  int32_t device, nray;
  double* ray_vec;       /* [nray][nstep_max+1][nv], zero past npoints */
  double* residual;      /* [nray][nstep_max+1] */
  int32_t* npoints;      /* [nray] */
  int32_t* stop_code;    /* [nray] */
  double* end_ray_vec;   /* [nray][nv] */
  double* end_residuals; /* [nray] */
  double* max_residuals; /* [nray] */
} rays_device_result_t;
int rays_hip_trace_gather(const rays_params_t* p, int nray, const double* rvec0, const double* rindex_vec0,
                          rays_device_result_t* result);
int rays_hip_result_to_host(const rays_params_t* p, const rays_device_result_t* result, double* ray_vec,
                            double* residual, int32_t* npoints, int32_t* stop_code, double* end_ray_vec,
                            double* end_residuals, double* max_residuals);

/* ---- multi-GPU trajectory exchange helpers ---------------------------------
 * The padded reference layout is mostly zeros (a ray uses npoints of nstep_max+1 slots).  Before
 * the RCCL gather a rank packs its slab to packed_vec[sum(npoints)][nv], packed_res[sum(npoints)]
 * (d_offsets = exclusive prefix sum of npoints, int64); the root unpacks a peer's block into its
 * slab of the padded global arrays (which must be zero-filled beforehand).  Device pointers,
 * asynchronous on hip_stream. */
int rays_hip_pack_device(int nray, int nv, int nstep_max, const int32_t* d_npoints,
                         const int64_t* d_offsets, const double* d_ray_vec, const double* d_residual,
                         double* d_packed_vec, double* d_packed_res, void* hip_stream);
int rays_hip_unpack_device(int nray, int nv, int nstep_max, const int32_t* d_npoints,
                           const int64_t* d_offsets, const double* d_packed_vec,
                           const double* d_packed_res, double* d_ray_vec, double* d_residual,
                           void* hip_stream);

/* ---- ray initialisation on the device (the step before the hot path) -------
 * Replaces the reference's serial launch loops
 *   ray_init_solovev_nphi_ntheta            solovev_ray_init_nphi_ntheta_m.f90:60-198
 *   ray_init_axisym_toroid_R_Z_nphi_ntheta  axisym_toroid_ray_init_R_Z_nphi_ntheta_m.f90:67-244
 *   simple_slab_ray_init                    simple_slab_ray_init_m.f90:59-187
 * (each: equilibrium at the launch point + cold dispersion root solve_n1_vs_n2_n3 per fan member,
 * dispersion_solvers_m.f90:49-112; evanescent launches are dropped and the survivors numbered in
 * loop order).  rays_fan_t mirrors the launcher's namelist
 * (/solovev_ray_init_nphi_ktheta_list/, /axisym_toroid_ray_init_R_Z_nphi_ntheta_list/,
 * /simple_slab_ray_init_list/) plus wave_mode / k0_sign of /rf_list/ (rf_m.f90:28-34). */
enum { RAYS_RAY_INIT_SOLOVEV_NPHI_NTHETA = 0, RAYS_RAY_INIT_AXISYM_R_Z_NPHI_NTHETA = 1,
       RAYS_RAY_INIT_SIMPLE_SLAB = 2 };
enum { RAYS_WAVE_PLUS = 0, RAYS_WAVE_MINUS = 1, RAYS_WAVE_FAST = 2, RAYS_WAVE_SLOW = 3 };
typedef struct rays_fan {
  int32_t model;      /* RAYS_RAY_INIT_* ; must match rays_params_t.equilib_model */
  int32_t wave_mode;  /* RAYS_WAVE_* */
  int32_t k0_sign;    /* +1 | -1 */
  /* solovev / axisym_toroid fans */
  int32_t n_r_launch, n_theta_launch; /* solovev: minor radius x poloidal angle; axisym: n_R x n_Z */
  int32_t n_rindex_theta, n_rindex_phi;
  double r_launch0, dr_launch, theta_launch0, dtheta_launch; /* axisym: r_launch0 = R, z_launch0 = Z */
  double z_launch0;
  double rindex_theta0, delta_rindex_theta, rindex_phi0, delta_rindex_phi;
  /* simple slab */
  int32_t n_x_launch, n_y_launch, n_z_launch, n_ky_launch, n_kz_launch, pad_;
  double x_launch0, dx_launch, y_launch0, dy_launch, slab_z_launch0;
  double rindex_y0, delta_rindex_y0, rindex_z0, delta_rindex_z0;
} rays_fan_t;
int rays_hip_sizeof_fan(void);

/* Host-pointer form: fills rvec0[nray_max][3], rindex_vec0[nray_max][3], ray_pwr_wt[nray_max]
 * (= the reference's allocatable module arrays of ray_init_m.f90:47-53, C order) and *nray.
 * Returns non-zero with the reference's message when the launcher would `stop 1`
 * (improper number of rays, no successful initialisation). */
int rays_hip_ray_init(const rays_params_t* p, const rays_fan_t* fan, int nray_max, double* rvec0,
                      double* rindex_vec0, double* ray_pwr_wt, int32_t* nray);
/* Device-pointer form (current device; d_* hold nray_max x 3 doubles): the fan never visits the
 * host.  Synchronises `hip_stream` once to return *nray. */
int rays_hip_ray_init_device(const rays_params_t* p, const rays_fan_t* fan, int nray_max,
                             double* d_rvec0, double* d_rindex_vec0, int32_t* nray, void* hip_stream);

/* ---- deposition profiles on the device (the step after the hot path) --------
 * Replaces calculate_deposition_profiles / bin_a_ray (post_process_lib/deposition_profiles_m.f90:
 * 228-292) with its evaluators Ptotal_axisym_psi / Ptotal_axisym_rho (:458-503) and the uniform
 * grid binner (math_functions_lib/bin_to_uniform_grid_m.f90: binner_real), for
 * equilib_model = 'axisym_toroid' runs with damping (nv >= 8), and the slab's Ptotal_x (evaluator
 * Ptotal_x_slab_evaluator :438-452, grid [xmin, xmax] of the slab box :134-137), applied to the
 * trajectory arrays where rays_hip_trace_device left them.
 *   d_work[n_bins][nray]   per-ray binned power: the reference's work(n_bins, nray), stored bin-major
 *                          (scratch of the call; transposed so that the reduction reads coalesce)
 *   d_profile_out[n_bins]  = d_profile_in (or 0) + sum over rays IN RAY ORDER, the order of the
 *                          reference's sum(work, 2): ranks that hold consecutive ray blocks chain
 *                          their partial sums through d_profile_in and obtain the single-process
 *                          result bit for bit (a few-KB exchange instead of the trajectory gather).
 * Grid = [0, 1] in psiN or rho, [xmin, xmax] for Ptotal_x; the reference's default is n_bins = 100
 * (n_bins <= 320 here). */
enum { RAYS_DEP_PTOTAL_PSI = 0, RAYS_DEP_PTOTAL_RHO = 1, RAYS_DEP_PTOTAL_X = 2 };
/* rho(psiN) spline of the eqdsk equilibrium (rho_profile of eqdsk_magnetics_spline_interp_m.f90:42,
 * 190-193; fspl(4, n) on grid(n)); needed for RAYS_DEP_PTOTAL_RHO.  Copied. */
int rays_hip_set_rho_table(const double* grid, const double* fspl, int n);
int rays_hip_deposition_device(const rays_params_t* p, int which, int n_bins, int nray,
                               const double* d_ray_vec, const int32_t* d_npoints,
                               const double* d_initial_ray_power, double* d_work,
                               const double* d_profile_in, double* d_profile_out, void* hip_stream);

/* Host-pointer form: ray_vec[nray][nstep_max+1][nv], npoints[nray], initial_ray_power[nray] are the
 * ray_results_m arrays; work[nray][n_bins] (= the reference's work(n_bins, nray), may be NULL) and
 * profile[n_bins] are filled.  Only points 1..maxval(npoints) of each ray cross PCIe. */
int rays_hip_deposition(const rays_params_t* p, int which, int n_bins, int nray, const double* ray_vec,
                        const int32_t* npoints, const double* initial_ray_power, double* work, double* profile);

/* The same profiles WITHOUT handing the trajectories back: the reference's drivers trace and post-process in one
 * process (RAYS_P.f90:19-44 -- trace_rays, then calculate_deposition_profiles on the same ray_results_m arrays,
 * deposition_profiles_m.f90:228-292).  rays_hip_keep_last_result(1) makes every later rays_hip_trace call leave the
 * device-resident image of its result (the padded ray_vec slabs and npoints of its blocks, on the devices that traced
 * them) in place until the next rays_hip_trace call, rays_hip_keep_last_result(0) or rays_hip_finalize; returns the
 * previous setting.  rays_hip_deposition_last bins that image: same arguments and results as rays_hip_deposition
 * minus the two arrays, blocks binned in ray order with the running sums carried from block to block (bit-identical
 * to the reference's ray-ordered sum for any number of blocks / devices).  Returns RAYS_HIP_NO_KEPT_RESULT (and
 * changes nothing) when no image of a trace with this nray / nv / nstep_max is held -- e.g. a post-processor that read
 * its arrays from a results file: call rays_hip_deposition then. */
#define RAYS_HIP_NO_KEPT_RESULT 5
int rays_hip_keep_last_result(int on);
int rays_hip_deposition_last(const rays_params_t* p, int which, int n_bins, int nray, const double* initial_ray_power,
                             double* work, double* profile);

/* Diagnostic entry used by the parity tests: evaluates equilibrium + deriv_cold + deriv_num +
 * eqn_ray + check_save at n states on the current device (host pointers; nv must be 7, nspec 1|2).
 * cold7/num7[n][7] = dddx(3) dddk(3) dddw; dvds[n][7]; resid[n]; codes[n][4] = equilibrium err,
 * eqn_ray stop code, check_save flag, check_save stop_ode. */
int rays_hip_probe(const rays_params_t* p, int n, const double* v, double* cold7, double* num7,
                   double* dvds, double* resid, int32_t* codes);

#ifdef __cplusplus
}
#endif
#endif /* RAYS_HIP_H */
