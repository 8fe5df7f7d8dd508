 module rays_hip_m
! iso_c_binding interface to librays_hip.so (include/rays_hip.h) -- the thin Fortran shim through
! which the RAYS host calls the MI355X ray-trajectory integrator.
!
! type(rays_params_t) mirrors the C struct field for field; the integer selectors are the images
! of the reference's string switches (see the enums in rays_hip.h).  Every real constant is passed
! by value from the host's own module variables (constants_m, rf_m, species_m ...), never
! re-derived on the device side: several of them are single-precision literals widened to double
! (constants_m.f90:39-48) and must reach the GPU bit for bit.

    use, intrinsic :: iso_c_binding
    implicit none

    integer(c_int), parameter :: RAYS_HIP_NO_KEPT_RESULT = 5   ! rays_hip_deposition_last: no device-resident image held

    integer(c_int), parameter :: RAYS_ABI_VERSION = 3
    integer, parameter :: RAYS_NS0 = 6   ! species_m nspec0 + 1

    ! selectors
    integer(c_int32_t), parameter :: RAYS_ODE_RK4 = 0, RAYS_ODE_SG = 1
    integer(c_int32_t), parameter :: RAYS_DERIV_COLD = 0, RAYS_DERIV_NUM = 1
    integer(c_int32_t), parameter :: RAYS_PARAM_ARCL = 0, RAYS_PARAM_TIME = 1
    integer(c_int32_t), parameter :: RAYS_EQ_SLAB = 0, RAYS_EQ_SOLOVEV = 1, RAYS_EQ_AXISYM = 2
    integer(c_int32_t), parameter :: RAYS_DAMP_NONE = 0, RAYS_DAMP_FUND_ECH = 1

    type, bind(C) :: rays_slab_params_t
        integer(c_int32_t) :: bx_prof_model, by_prof_model, bz_prof_model, dens_prof_model
        integer(c_int32_t) :: t_prof_model(RAYS_NS0)
        integer(c_int32_t) :: pad_(2)
        real(c_double) :: xmin, xmax, ymin, ymax, zmin, zmax
        real(c_double) :: rmaj, rmin, x0
        real(c_double) :: bx0, by0, bz0, LBy_shear_scale, LBz_scale, dBzdx
        real(c_double) :: Ln_scale, dndx, alphan1, alphan2, n_min
        real(c_double) :: LT_scale, dtdx
        real(c_double) :: alphat1(RAYS_NS0), alphat2(RAYS_NS0), T_min(RAYS_NS0)
    end type rays_slab_params_t

    type, bind(C) :: rays_solovev_params_t
        integer(c_int32_t) :: dens_prof_model
        integer(c_int32_t) :: t_prof_model(RAYS_NS0)
        integer(c_int32_t) :: pad_(1)
        real(c_double) :: rmaj, kappa, bphi0, iota0, outer_bound
        real(c_double) :: psiB
        real(c_double) :: alphan1, alphan2
        real(c_double) :: alphat1(RAYS_NS0), alphat2(RAYS_NS0)
        real(c_double) :: box_rmin, box_rmax, box_zmin, box_zmax
    end type rays_solovev_params_t

    type, bind(C) :: rays_axisym_params_t
        integer(c_int32_t) :: magnetics_model, density_prof_model
        integer(c_int32_t) :: t_prof_model(RAYS_NS0)
        real(c_double) :: box_rmin, box_rmax, box_zmin, box_zmax
        real(c_double) :: plasma_psi_limit
        real(c_double) :: psiB
        real(c_double) :: alphan1, alphan2, d_scrape_off, T_scrape_off
        real(c_double) :: alphat1(RAYS_NS0), alphat2(RAYS_NS0)
    end type rays_axisym_params_t

    ! spline tables of the eqdsk equilibrium: pointers to the host's own (contiguous) arrays
    type, bind(C) :: rays_axisym_tables_t
        integer(c_int32_t) :: nr, nz, n_rb, n_ne, n_te, n_ti
        type(c_ptr) :: r_grid, z_grid, psi_fspl
        type(c_ptr) :: rb_grid, rb_fspl
        type(c_ptr) :: ne_grid, ne_fspl
        type(c_ptr) :: te_grid, te_fspl
        type(c_ptr) :: ti_grid, ti_fspl
    end type rays_axisym_tables_t

    type, bind(C) :: rays_params_t
        integer(c_int32_t) :: abi_version
        integer(c_int32_t) :: nv, nspec, nstep_max
        integer(c_int32_t) :: ode_solver, ray_deriv, ray_param, equilib_model
        integer(c_int32_t) :: integrate_eq_gradients
        integer(c_int32_t) :: pad_(3)
        real(c_double) :: ds, s_max
        real(c_double) :: omgrf, k0
        real(c_double) :: clight, eps0
        real(c_double) :: dispersion_resid_limit
        real(c_double) :: rel_err0, abs_err0, SG_error_limit
        real(c_double) :: qs(RAYS_NS0), ms(RAYS_NS0)
        real(c_double) :: n0s(RAYS_NS0), t0s(RAYS_NS0)
        real(c_double) :: eta(RAYS_NS0)
        type(rays_slab_params_t) :: slab
        type(rays_solovev_params_t) :: solovev
        integer(c_int32_t) :: damping_model, multi_spec_damping
        real(c_double) :: total_damping_limit
        type(rays_axisym_params_t) :: axisym
    end type rays_params_t

    ! ray launchers on the device (rays_fan_t of rays_hip.h; SURVEY 8(f) f1)
    integer(c_int32_t), parameter :: RAYS_RAY_INIT_SOLOVEV_NPHI_NTHETA = 0, &
         & RAYS_RAY_INIT_AXISYM_R_Z_NPHI_NTHETA = 1, RAYS_RAY_INIT_SIMPLE_SLAB = 2
    integer(c_int32_t), parameter :: RAYS_WAVE_PLUS = 0, RAYS_WAVE_MINUS = 1, RAYS_WAVE_FAST = 2, RAYS_WAVE_SLOW = 3
    type, bind(C) :: rays_fan_t
       integer(c_int32_t) :: model, wave_mode, k0_sign
       integer(c_int32_t) :: n_r_launch, n_theta_launch, n_rindex_theta, n_rindex_phi
       real(c_double) :: r_launch0, dr_launch, theta_launch0, dtheta_launch, z_launch0
       real(c_double) :: rindex_theta0, delta_rindex_theta, rindex_phi0, delta_rindex_phi
       integer(c_int32_t) :: n_x_launch, n_y_launch, n_z_launch, n_ky_launch, n_kz_launch, pad_
       real(c_double) :: x_launch0, dx_launch, y_launch0, dy_launch, slab_z_launch0
       real(c_double) :: rindex_y0, delta_rindex_y0, rindex_z0, delta_rindex_z0
    end type rays_fan_t

    ! device-resident result of rays_hip_trace_gather (rays_device_result_t of rays_hip.h)
    type, bind(C) :: rays_device_result_t
       integer(c_int32_t) :: device, nray
       type(c_ptr) :: ray_vec, residual, npoints, stop_code, end_ray_vec, end_residuals, max_residuals
    end type rays_device_result_t

    integer(c_int), parameter :: RAYS_TRACE_NO_ZERO_FILL = 1
    integer(c_int), parameter :: RAYS_NUMERICS_EXACT = 0, RAYS_NUMERICS_TOLERANCE = 1
    integer(c_int), parameter :: RAYS_DEP_PTOTAL_PSI = 0, RAYS_DEP_PTOTAL_RHO = 1, RAYS_DEP_PTOTAL_X = 2

    interface

       integer(c_int) function rays_hip_init(ngpu) bind(C, name='rays_hip_init')
          import :: c_int
          integer(c_int), value :: ngpu
       end function rays_hip_init

       ! explicit slot -> device list for rays_hip_trace / rays_hip_trace_gather
       integer(c_int) function rays_hip_init_devices(n, device_ids) bind(C, name='rays_hip_init_devices')
          import :: c_int
          integer(c_int), value :: n
          integer(c_int), intent(in) :: device_ids(*)
       end function rays_hip_init_devices

       integer(c_int) function rays_hip_finalize() bind(C, name='rays_hip_finalize')
          import :: c_int
       end function rays_hip_finalize

       ! numerics of the trace kernels: RAYS_NUMERICS_EXACT (0, default: bit-identical to the CPU path) or
       ! RAYS_NUMERICS_TOLERANCE (1: every step within 1e-10 relative, counts and stop flags exact; cold RK4 kernels)
       integer(c_int) function rays_hip_set_numerics(mode) bind(C, name='rays_hip_set_numerics')
          import :: c_int
          integer(c_int), value :: mode
       end function rays_hip_set_numerics

       integer(c_int) function rays_hip_get_numerics() bind(C, name='rays_hip_get_numerics')
          import :: c_int
       end function rays_hip_get_numerics

       integer(c_int) function rays_hip_last_error(buf, len) bind(C, name='rays_hip_last_error')
          import :: c_int, c_char
          character(kind=c_char) :: buf(*)
          integer(c_int), value :: len
       end function rays_hip_last_error

       type(c_ptr) function rays_hip_stop_flag_text(stop_code) bind(C, name='rays_hip_stop_flag_text')
          import :: c_int, c_ptr
          integer(c_int), value :: stop_code
       end function rays_hip_stop_flag_text

       integer(c_int) function rays_hip_set_zfun_table(fspl_re, nx, x_min, x_max) &
                    & bind(C, name='rays_hip_set_zfun_table')
          import :: c_int, c_double
          real(c_double), intent(in) :: fspl_re(4,*)
          integer(c_int), value :: nx
          real(c_double), value :: x_min, x_max
       end function rays_hip_set_zfun_table

       integer(c_int) function rays_hip_set_axisym_tables(t) bind(C, name='rays_hip_set_axisym_tables')
          import :: c_int, rays_axisym_tables_t
          type(rays_axisym_tables_t), intent(in) :: t
       end function rays_hip_set_axisym_tables

       ! magnetics_model = 'eqdsk_magnetics_lin_interp': eqdsk_utilities_m's R_grid, Z_grid, Psi, T (rb_fspl), dR, dZ
       integer(c_int) function rays_hip_set_eqdsk_lin_tables(t, dR, dZ) bind(C, name='rays_hip_set_eqdsk_lin_tables')
          import :: c_int, c_double, rays_axisym_tables_t
          type(rays_axisym_tables_t), intent(in) :: t
          real(c_double), value :: dR, dZ
       end function rays_hip_set_eqdsk_lin_tables

       integer(c_int) function rays_hip_check_params(p) bind(C, name='rays_hip_check_params')
          import :: c_int, rays_params_t
          type(rays_params_t), intent(in) :: p
       end function rays_hip_check_params

       ! Replaces `call trace_rays`: blocking, host arrays in the reference's own layouts.
       integer(c_int) function rays_hip_trace(p, nray, rvec0, rindex_vec0, ray_vec, residual, &
                    & npoints, stop_code, end_ray_vec, end_residuals, max_residuals, elapsed_s) &
                    & bind(C, name='rays_hip_trace')
          import :: c_int, c_int32_t, c_double, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: nray
          real(c_double), intent(in) :: rvec0(3,*), rindex_vec0(3,*)
          real(c_double), intent(inout) :: ray_vec(*), residual(*)
          integer(c_int32_t), intent(inout) :: npoints(*), stop_code(*)
          real(c_double), intent(inout) :: end_ray_vec(*), end_residuals(*), max_residuals(*)
          real(c_double), intent(out) :: elapsed_s
       end function rays_hip_trace

       ! Multi-GPU trace whose result stays in device memory on the root device (RCCL gather of the blocks'
       ! packed trajectories: SURVEY 8(e)); rays_hip_result_to_host copies it into ray_results_m's arrays,
       ! rays_hip_deposition_device consumes it in place.
       integer(c_int) function rays_hip_trace_gather(p, nray, rvec0, rindex_vec0, result) &
                    & bind(C, name='rays_hip_trace_gather')
          import :: c_int, c_double, rays_params_t, rays_device_result_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: nray
          real(c_double), intent(in) :: rvec0(3,*), rindex_vec0(3,*)
          type(rays_device_result_t), intent(out) :: result
       end function rays_hip_trace_gather

       integer(c_int) function rays_hip_result_to_host(p, result, ray_vec, residual, npoints, stop_code, &
                    & end_ray_vec, end_residuals, max_residuals) bind(C, name='rays_hip_result_to_host')
          import :: c_int, c_int32_t, c_double, rays_params_t, rays_device_result_t
          type(rays_params_t), intent(in) :: p
          type(rays_device_result_t), intent(in) :: result
          real(c_double), intent(inout) :: ray_vec(*), residual(*)
          integer(c_int32_t), intent(inout) :: npoints(*), stop_code(*)
          real(c_double), intent(inout) :: end_ray_vec(*), end_residuals(*), max_residuals(*)
       end function rays_hip_result_to_host

       ! Device-pointer forms (type(c_ptr) = device memory on the current HIP device; hip_stream = a hipStream_t
       ! or c_null_ptr): the trace, the fused `ds` scan (ray_scan.f90:33-49), one ode_solver step from arbitrary
       ! states (ode_m.f90:218-254), the device ray launcher and the deposition profiles.
       integer(c_int) function rays_hip_trace_device(p, nray, d_rvec0, d_rindex_vec0, d_ray_vec, d_residual, &
                    & d_npoints, d_stop_code, d_end_ray_vec, d_end_residuals, d_max_residuals, hip_stream, flags) &
                    & bind(C, name='rays_hip_trace_device')
          import :: c_int, c_ptr, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: nray
          type(c_ptr), value :: d_rvec0, d_rindex_vec0, d_ray_vec, d_residual, d_npoints, d_stop_code
          type(c_ptr), value :: d_end_ray_vec, d_end_residuals, d_max_residuals, hip_stream
          integer(c_int), value :: flags
       end function rays_hip_trace_device

       integer(c_int) function rays_hip_scan_device(p, n_runs, d_ds_values, nray, d_rvec0, d_rindex_vec0, &
                    & d_ray_vec, d_residual, d_npoints, d_stop_code, d_end_ray_vec, d_end_residuals, &
                    & d_max_residuals, hip_stream, flags) bind(C, name='rays_hip_scan_device')
          import :: c_int, c_ptr, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: n_runs, nray
          type(c_ptr), value :: d_ds_values, d_rvec0, d_rindex_vec0, d_ray_vec, d_residual, d_npoints, d_stop_code
          type(c_ptr), value :: d_end_ray_vec, d_end_residuals, d_max_residuals, hip_stream
          integer(c_int), value :: flags
       end function rays_hip_scan_device

       ! ASYNCHRONOUS on hip_stream (blocking before round 3): synchronise the stream before d_v1 / d_resid /
       ! d_stop_code are read on the host or from another stream; one caller thread per (device, stream).
       integer(c_int) function rays_hip_ode_step_device(p, n, d_v0, d_s0, d_v1, d_resid, d_stop_code, hip_stream) &
                    & bind(C, name='rays_hip_ode_step_device')
          import :: c_int, c_ptr, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: n
          type(c_ptr), value :: d_v0, d_s0, d_v1, d_resid, d_stop_code, hip_stream
       end function rays_hip_ode_step_device

       integer(c_int) function rays_hip_ray_init_device(p, fan, nray_max, d_rvec0, d_rindex_vec0, nray, hip_stream) &
                    & bind(C, name='rays_hip_ray_init_device')
          import :: c_int, c_int32_t, c_ptr, rays_params_t, rays_fan_t
          type(rays_params_t), intent(in) :: p
          type(rays_fan_t), intent(in) :: fan
          integer(c_int), value :: nray_max
          type(c_ptr), value :: d_rvec0, d_rindex_vec0, hip_stream
          integer(c_int32_t), intent(out) :: nray
       end function rays_hip_ray_init_device

       ! Deposition profiles (post_process_lib/deposition_profiles_m.f90:228-292).  rho(psiN) spline for 'Ptotal_rho':
       integer(c_int) function rays_hip_set_rho_table(grid, fspl, n) bind(C, name='rays_hip_set_rho_table')
          import :: c_int, c_double
          real(c_double), intent(in) :: grid(*), fspl(4,*)
          integer(c_int), value :: n
       end function rays_hip_set_rho_table

       integer(c_int) function rays_hip_deposition_device(p, which, n_bins, nray, d_ray_vec, d_npoints, &
                    & d_initial_ray_power, d_work, d_profile_in, d_profile_out, hip_stream) &
                    & bind(C, name='rays_hip_deposition_device')
          import :: c_int, c_ptr, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: which, n_bins, nray
          type(c_ptr), value :: d_ray_vec, d_npoints, d_initial_ray_power, d_work, d_profile_in, d_profile_out, hip_stream
       end function rays_hip_deposition_device

       ! host arrays in the reference's layouts: ray_vec(nv, nstep_max+1, nray), work(n_bins, nray), profile(n_bins)
       integer(c_int) function rays_hip_deposition(p, which, n_bins, nray, ray_vec, npoints, initial_ray_power, &
                    & work, profile) bind(C, name='rays_hip_deposition')
          import :: c_int, c_int32_t, c_double, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: which, n_bins, nray
          real(c_double), intent(in) :: ray_vec(*), initial_ray_power(*)
          integer(c_int32_t), intent(in) :: npoints(*)
          real(c_double), intent(inout) :: work(*), profile(*)
       end function rays_hip_deposition

       ! The same profiles of the rays the LAST rays_hip_trace call traced, binned from the image that call left on
       ! the device(s) (rays_hip_keep_last_result(1) before the trace): no trajectory upload.  Returns
       ! RAYS_HIP_NO_KEPT_RESULT when no image of a trace with this nray / nv / nstep_max is held.
       integer(c_int) function rays_hip_keep_last_result(on) bind(C, name='rays_hip_keep_last_result')
          import :: c_int
          integer(c_int), value :: on
       end function rays_hip_keep_last_result
       integer(c_int) function rays_hip_deposition_last(p, which, n_bins, nray, initial_ray_power, work, profile) &
                    & bind(C, name='rays_hip_deposition_last')
          import :: c_int, c_double, rays_params_t
          type(rays_params_t), intent(in) :: p
          integer(c_int), value :: which, n_bins, nray
          real(c_double), intent(in) :: initial_ray_power(*)
          real(c_double), intent(inout) :: work(*), profile(*)
       end function rays_hip_deposition_last

       ! Replaces the serial launch loops of ray_init_m's launchers (solovev_ray_init_nphi_ntheta_m.f90:
       ! 60-198 etc.): fills rvec0(3,nray_max), rindex_vec0(3,nray_max), ray_pwr_wt(nray_max), nray.
       integer(c_int) function rays_hip_ray_init(p, fan, nray_max, rvec0, rindex_vec0, ray_pwr_wt, nray) &
                    & bind(C, name='rays_hip_ray_init')
          import :: c_int, c_int32_t, c_double, rays_params_t, rays_fan_t
          type(rays_params_t), intent(in) :: p
          type(rays_fan_t), intent(in) :: fan
          integer(c_int), value :: nray_max
          real(c_double), intent(inout) :: rvec0(3,*), rindex_vec0(3,*), ray_pwr_wt(*)
          integer(c_int32_t), intent(out) :: nray
       end function rays_hip_ray_init

    end interface

 contains

    integer(c_int32_t) function wave_mode_code(wave_mode)
    ! rf_m's wave_mode string -> RAYS_WAVE_* (dispersion_solvers_m.f90:86-101 stops on anything else)
       character(len=*), intent(in) :: wave_mode
       select case (trim(wave_mode))
          case ('plus');  wave_mode_code = RAYS_WAVE_PLUS
          case ('minus'); wave_mode_code = RAYS_WAVE_MINUS
          case ('fast');  wave_mode_code = RAYS_WAVE_FAST
          case ('slow');  wave_mode_code = RAYS_WAVE_SLOW
          case default
             write(0,*) 'solve_disp: improper wave_mode = ', trim(wave_mode); stop 1
       end select
    end function wave_mode_code

    subroutine clear_fan(fan)
       type(rays_fan_t), intent(out) :: fan
       fan%model = 0 ; fan%wave_mode = 0 ; fan%k0_sign = 1
       fan%n_r_launch = 0 ; fan%n_theta_launch = 0 ; fan%n_rindex_theta = 0 ; fan%n_rindex_phi = 0
       fan%r_launch0 = 0. ; fan%dr_launch = 0. ; fan%theta_launch0 = 0. ; fan%dtheta_launch = 0. ; fan%z_launch0 = 0.
       fan%rindex_theta0 = 0. ; fan%delta_rindex_theta = 0. ; fan%rindex_phi0 = 0. ; fan%delta_rindex_phi = 0.
       fan%n_x_launch = 0 ; fan%n_y_launch = 0 ; fan%n_z_launch = 0 ; fan%n_ky_launch = 0
       fan%n_kz_launch = 0 ; fan%pad_ = 0
       fan%x_launch0 = 0. ; fan%dx_launch = 0. ; fan%y_launch0 = 0. ; fan%dy_launch = 0.
       fan%slab_z_launch0 = 0. ; fan%rindex_y0 = 0. ; fan%delta_rindex_y0 = 0.
       fan%rindex_z0 = 0. ; fan%delta_rindex_z0 = 0.
    end subroutine clear_fan

    function stop_flag_string(stop_code) result(flag)
    ! integer stop code -> the reference's ode_stop_flag text (e.g. ' nstep > nstep_max')
       integer(c_int), intent(in) :: stop_code
       character(len=60) :: flag
       type(c_ptr) :: cp
       character(kind=c_char), pointer :: cs(:)
       integer :: i
       flag = ''
       cp = rays_hip_stop_flag_text(stop_code)
       if (.not. c_associated(cp)) return
       call c_f_pointer(cp, cs, [60])
       do i = 1, 60
          if (cs(i) == c_null_char) exit
          flag(i:i) = cs(i)
       end do
    end function stop_flag_string

    subroutine last_error_string(msg)
       character(len=*), intent(out) :: msg
       character(kind=c_char) :: buf(512)
       integer :: i, n
       msg = ''
       n = rays_hip_last_error(buf, 512_c_int)
       do i = 1, min(len(msg), 511)
          if (buf(i) == c_null_char) exit
          msg(i:i) = buf(i)
       end do
    end subroutine last_error_string

 end module rays_hip_m
