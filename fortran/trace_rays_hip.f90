 subroutine trace_rays
! Drop-in replacement for RAYS_project/RAYS_lib/ray_tracing.f90 (`subroutine trace_rays`).
!
! Same contract: reads the module state left by initialize(), fills the ray_results_m arrays.
! Instead of the OpenMP ray loop it packs the module state into type(rays_params_t) and makes ONE
! blocking call into librays_hip.so (rays_hip_trace), which shards the rays over the node's
! MI355X GPUs.  Configuration errors the device path does not support stop with the library's
! message, like the reference's own `stop 1`.
!
! Build: compile this file INSTEAD of ray_tracing.f90, together with rays_hip_m.f90, and link
! -lrays_hip (see INTEGRATION.md).

    use, intrinsic :: iso_c_binding
    use constants_m, only : rkind, clight, eps0
    use diagnostics_m, only : message, text_message, integrate_eq_gradients
    use species_m, only : nspec, qs, ms, n0s, t0s, eta
    use rf_m, only : omgrf, k0, ray_param, ray_dispersion_model, dispersion_resid_limit
    use damping_m, only : damping_model, multi_spec_damping, total_damping_limit
    use zfunctions_m, only : fsplRe, zf_nx => nx, x_grid_min, x_grid_max
    use equilibrium_m, only : equilib_model
    use slab_eq_m, only : s_xmin => xmin, s_xmax => xmax, s_ymin => ymin, s_ymax => ymax, &
         & s_zmin => zmin, s_zmax => zmax, s_rmaj => rmaj, s_rmin => rmin, s_x0 => x0, &
         & bx_prof_model, by_prof_model, bz_prof_model, bx0, by0, bz0, LBy_shear_scale, LBz_scale, &
         & dBzdx, s_dens => dens_prof_model, Ln_scale, dndx, s_alphan1 => alphan1, &
         & s_alphan2 => alphan2, n_min, s_tmodel => t_prof_model, LT_scale, dtdx, &
         & s_alphat1 => alphat1, s_alphat2 => alphat2, T_min
    use solovev_eq_m, only : v_rmaj => rmaj, kappa, bphi0, iota0, outer_bound, psiB, &
         & v_dens => dens_prof_model, v_alphan1 => alphan1, v_alphan2 => alphan2, &
         & v_tmodel => t_prof_model, v_alphat1 => alphat1, v_alphat2 => alphat2, &
         & box_rmin, box_rmax, box_zmin, box_zmax
    use axisym_toroid_eq_m, only : magnetics_model, a_dens => density_prof_model, a_tmodel => temperature_prof_model, &
         & a_rmin => box_rmin, a_rmax => box_rmax, a_zmin => box_zmin, a_zmax => box_zmax, plasma_psi_limit, &
         & a_alphan1 => alphan1, a_alphan2 => alphan2, d_scrape_off, T_scrape_off, &
         & a_alphat1 => alphat1, a_alphat2 => alphat2
    use eqdsk_magnetics_spline_interp_m, only : Psi_profile, T_profile
    use eqdsk_utilities_m, only : PSIBOUND
    use density_spline_interp_m, only : ne_profile_N
    use temperature_spline_interp_m, only : Te_profileN, Ti_profileN
    use ode_m, only : nv, ds, s_max, nstep_max, ode_solver_name, ray_deriv_name
    use ray_init_m, only : nray, rvec0, rindex_vec0, ray_pwr_wt
    use ray_results_m, only : ray_stop_flag, ray_vec, residual, npoints, end_residuals, &
         & max_residuals, end_ray_parameter, start_ray_vec, end_ray_vec, initial_ray_power, &
         & ray_trace_time, total_trace_time
    use rays_hip_m

    implicit none

    type(rays_params_t) :: p
    type(rays_axisym_tables_t) :: tab
    ! contiguous TARGET copies of the host's spline objects (c_loc needs a target)
    real(c_double), allocatable, target :: t_rg(:), t_zg(:), t_psi(:,:,:,:), t_rbg(:), t_rb(:,:)
    real(c_double), allocatable, target :: t_neg(:), t_ne(:,:), t_teg(:), t_te(:,:), t_tig(:), t_ti(:,:)
    integer(c_int32_t), allocatable :: stop_code(:)
    integer(c_int) :: rc
    real(c_double) :: elapsed
    character(len=512) :: msg
    integer :: iray, is
    ! SG_ode_m is a submodule of ode_m: its namelist variables are not use-associable, so the
    ! shim reads /SG_ode_list/ itself (same file, same group, same defaults: SG_ode_m.f90:26-31).
    real(KIND=rkind) :: rel_err0, abs_err0, SG_error_limit
    namelist /SG_ode_list/ rel_err0, abs_err0, SG_error_limit
    integer :: input_unit, get_unit_number, ios
    logical :: is_open

    rel_err0 = 0. ; abs_err0 = 0. ; SG_error_limit = 0.1
    if (trim(ode_solver_name) == 'SG_ODE') then
       ! the reference can leave 'rays.in' connected (openmp_m.f90:53-57 jumps over its close)
       inquire(file='rays.in', opened=is_open, number=input_unit)
       if (is_open) then
          rewind(input_unit)
          read(input_unit, SG_ode_list, iostat=ios)
       else
          input_unit = get_unit_number()
          open(unit=input_unit, file='rays.in', action='read', status='old', form='formatted')
          read(input_unit, SG_ode_list, iostat=ios)
          close(unit=input_unit)
       end if
    end if

    p%abi_version = RAYS_ABI_VERSION
    p%nv = nv
    p%nspec = nspec
    p%nstep_max = nstep_max
    p%pad_ = 0
    p%integrate_eq_gradients = merge(1, 0, integrate_eq_gradients)

    select case (trim(ode_solver_name))
       case ('RK4_ODE'); p%ode_solver = RAYS_ODE_RK4
       case ('SG_ODE');  p%ode_solver = RAYS_ODE_SG
       case default
          write(0,*) 'ode_solver, invalid ode solver = ', trim(ode_solver_name); stop 2
    end select
    select case (trim(ray_deriv_name))
       case ('cold');      p%ray_deriv = RAYS_DERIV_COLD
       case ('numerical'); p%ray_deriv = RAYS_DERIV_NUM
       case default
          write(*,*) 'EQN_RAY: invalid value, ray_deriv_name = ', ray_deriv_name; stop 1
    end select
    select case (ray_param)
       case ('arcl'); p%ray_param = RAYS_PARAM_ARCL
       case ('time'); p%ray_param = RAYS_PARAM_TIME
       case default
          write(0,*) 'EQN_RAY: invalid ray parameter = ', ray_param; stop 1
    end select
    if (trim(ray_dispersion_model) /= 'cold') stop 'check_save: unimplemented ray_dispersion_model'
    select case (trim(damping_model))
       case ('no_damp');       p%damping_model = RAYS_DAMP_NONE
       case ('damp_fund_ECH'); p%damping_model = RAYS_DAMP_FUND_ECH
          ! the Z-function spline table built by initialize_damping_m (zfunctions_m.f90:436-466)
          if (rays_hip_set_zfun_table(fsplRe, int(zf_nx, c_int), x_grid_min, x_grid_max) /= 0) then
             call last_error_string(msg)
             write(0,*) 'trace_rays (HIP): ', trim(msg) ; stop 1
          end if
       case default
          write (*, *) 'damping: Unimplemented damping model ', trim(damping_model); stop 1
    end select
    p%multi_spec_damping = merge(1, 0, multi_spec_damping)
    p%total_damping_limit = total_damping_limit

    p%ds = ds ; p%s_max = s_max
    p%omgrf = omgrf ; p%k0 = k0 ; p%clight = clight ; p%eps0 = eps0
    p%dispersion_resid_limit = dispersion_resid_limit
    p%rel_err0 = rel_err0 ; p%abs_err0 = abs_err0 ; p%SG_error_limit = SG_error_limit
    p%qs = qs(0:5) ; p%ms = ms(0:5) ; p%n0s = n0s(0:5) ; p%t0s = t0s(0:5) ; p%eta = eta(0:5)

    ! zero both equilibrium blocks, then fill the active one
    p%slab%t_prof_model = 0 ; p%slab%pad_ = 0
    p%slab%alphat1 = 0. ; p%slab%alphat2 = 0. ; p%slab%T_min = 0.
    p%solovev%t_prof_model = 0 ; p%solovev%pad_ = 0
    p%solovev%alphat1 = 0. ; p%solovev%alphat2 = 0.

    select case (trim(equilib_model))
    case ('slab')
       p%equilib_model = RAYS_EQ_SLAB
       p%slab%bx_prof_model = pick(bx_prof_model, [character(len=12) :: 'zero'])
       p%slab%by_prof_model = pick(by_prof_model, [character(len=12) :: 'zero', 'constant', 'toroid', 'linear_shear'])
       p%slab%bz_prof_model = pick(bz_prof_model, [character(len=12) :: 'constant', 'toroid', 'linear', 'linear_2'])
       p%slab%dens_prof_model = pick(s_dens, [character(len=12) :: 'constant', 'linear', 'linear_2', 'parabolic', 'Gaussian'])
       do is = 0, nspec
          p%slab%t_prof_model(is+1) = pick(s_tmodel(is), [character(len=12) :: 'zero', 'constant', 'linear', 'linear_2', 'parabolic'])
          p%slab%alphat1(is+1) = s_alphat1(is) ; p%slab%alphat2(is+1) = s_alphat2(is)
          p%slab%T_min(is+1) = T_min(is)
       end do
       p%slab%xmin = s_xmin ; p%slab%xmax = s_xmax ; p%slab%ymin = s_ymin ; p%slab%ymax = s_ymax
       p%slab%zmin = s_zmin ; p%slab%zmax = s_zmax
       p%slab%rmaj = s_rmaj ; p%slab%rmin = s_rmin ; p%slab%x0 = s_x0
       p%slab%bx0 = bx0 ; p%slab%by0 = by0 ; p%slab%bz0 = bz0
       p%slab%LBy_shear_scale = LBy_shear_scale ; p%slab%LBz_scale = LBz_scale ; p%slab%dBzdx = dBzdx
       p%slab%Ln_scale = Ln_scale ; p%slab%dndx = dndx
       p%slab%alphan1 = s_alphan1 ; p%slab%alphan2 = s_alphan2 ; p%slab%n_min = n_min
       p%slab%LT_scale = LT_scale ; p%slab%dtdx = dtdx
    case ('solovev')
       p%equilib_model = RAYS_EQ_SOLOVEV
       p%solovev%dens_prof_model = pick(v_dens, [character(len=12) :: 'constant', 'parabolic'])
       do is = 0, nspec
          ! 'zero' -> 0, 'parabolic' -> 2 ('constant' leaves ts undefined in the reference: rejected)
          p%solovev%t_prof_model(is+1) = pick(v_tmodel(is), [character(len=12) :: 'zero', '?', 'parabolic'])
          p%solovev%alphat1(is+1) = v_alphat1(is) ; p%solovev%alphat2(is+1) = v_alphat2(is)
       end do
       p%solovev%rmaj = v_rmaj ; p%solovev%kappa = kappa ; p%solovev%bphi0 = bphi0
       p%solovev%iota0 = iota0 ; p%solovev%outer_bound = outer_bound ; p%solovev%psiB = psiB
       p%solovev%alphan1 = v_alphan1 ; p%solovev%alphan2 = v_alphan2
       p%solovev%box_rmin = box_rmin ; p%solovev%box_rmax = box_rmax
       p%solovev%box_zmin = box_zmin ; p%solovev%box_zmax = box_zmax
    case ('axisym_toroid')
       p%equilib_model = RAYS_EQ_AXISYM
       p%axisym%magnetics_model = pick(magnetics_model, [character(len=32) :: 'eqdsk_magnetics_spline_interp'])
       p%axisym%density_prof_model = pick(a_dens, [character(len=32) :: 'constant', 'parabolic', 'density_spline_interp'])
       p%axisym%t_prof_model = 0 ; p%axisym%alphat1 = 0. ; p%axisym%alphat2 = 0.
       do is = 0, nspec
          p%axisym%t_prof_model(is+1) = pick(a_tmodel(is), [character(len=32) :: 'zero', 'constant', 'parabolic', &
               & 'temperature_spline_interp'])
          p%axisym%alphat1(is+1) = a_alphat1(is) ; p%axisym%alphat2(is+1) = a_alphat2(is)
       end do
       p%axisym%box_rmin = a_rmin ; p%axisym%box_rmax = a_rmax
       p%axisym%box_zmin = a_zmin ; p%axisym%box_zmax = a_zmax
       p%axisym%plasma_psi_limit = plasma_psi_limit
       p%axisym%psiB = PSIBOUND     ! already PSIBOUND - PSIAXIS (eqdsk_magnetics_spline_interp_m.f90:172)
       p%axisym%alphan1 = a_alphan1 ; p%axisym%alphan2 = a_alphan2
       p%axisym%d_scrape_off = d_scrape_off ; p%axisym%T_scrape_off = T_scrape_off
       ! spline tables built by initialize_eqdsk_magnetics_spline_interp / initialize_*_spline_interp
       t_rg = Psi_profile%x_grid ; t_zg = Psi_profile%y_grid ; t_psi = Psi_profile%fspl
       t_rbg = T_profile%x_grid ; t_rb = T_profile%fspl
       tab%nr = Psi_profile%nx ; tab%nz = Psi_profile%ny ; tab%n_rb = T_profile%nx
       tab%r_grid = c_loc(t_rg) ; tab%z_grid = c_loc(t_zg) ; tab%psi_fspl = c_loc(t_psi)
       tab%rb_grid = c_loc(t_rbg) ; tab%rb_fspl = c_loc(t_rb)
       tab%n_ne = 0 ; tab%n_te = 0 ; tab%n_ti = 0
       tab%ne_grid = c_null_ptr ; tab%ne_fspl = c_null_ptr
       tab%te_grid = c_null_ptr ; tab%te_fspl = c_null_ptr
       tab%ti_grid = c_null_ptr ; tab%ti_fspl = c_null_ptr
       if (allocated(ne_profile_N%fspl)) then
          t_neg = ne_profile_N%x_grid ; t_ne = ne_profile_N%fspl
          tab%n_ne = ne_profile_N%nx ; tab%ne_grid = c_loc(t_neg) ; tab%ne_fspl = c_loc(t_ne)
       end if
       if (allocated(Te_profileN%fspl)) then
          t_teg = Te_profileN%x_grid ; t_te = Te_profileN%fspl
          tab%n_te = Te_profileN%nx ; tab%te_grid = c_loc(t_teg) ; tab%te_fspl = c_loc(t_te)
       end if
       if (allocated(Ti_profileN%fspl)) then
          t_tig = Ti_profileN%x_grid ; t_ti = Ti_profileN%fspl
          tab%n_ti = Ti_profileN%nx ; tab%ti_grid = c_loc(t_tig) ; tab%ti_fspl = c_loc(t_ti)
       end if
       if (rays_hip_set_axisym_tables(tab) /= 0) then
          call last_error_string(msg)
          write(0,*) 'trace_rays (HIP): ', trim(msg) ; stop 1
       end if
    case default
       write(0,*) 'trace_rays (HIP): equilib_model not on the device path = ', trim(equilib_model); stop 1
    end select

    if (rays_hip_check_params(p) /= 0) then
       call last_error_string(msg)
       write(0,*) 'trace_rays (HIP): ', trim(msg) ; stop 1
    end if

    allocate(stop_code(nray))
    stop_code = 0
    ray_stop_flag = ''

    rc = rays_hip_trace(p, int(nray, c_int), rvec0, rindex_vec0, ray_vec, residual, npoints, &
         & stop_code, end_ray_vec, end_residuals, max_residuals, elapsed)
    if (rc /= 0) then
       call last_error_string(msg)
       write(0,*) 'trace_rays (HIP): rays_hip_trace failed: ', trim(msg) ; stop 1
    end if

    ! per-ray summary fields of ray_results_m (ray_tracing.f90:252-260)
    do iray = 1, nray
       ray_stop_flag(iray) = stop_flag_string(stop_code(iray))
       ! rays that did not start keep their zero-initialised summary fields (ray_tracing.f90:101-112)
       if (npoints(iray) == 1 .and. all(end_ray_vec(:,iray) == 0.)) cycle
       initial_ray_power(iray) = ray_pwr_wt(iray)
       ray_trace_time(iray) = elapsed/nray
       end_ray_parameter(iray) = end_ray_vec(7,iray)
       start_ray_vec(:,iray) = ray_vec(:,1,iray)
    end do
    total_trace_time = elapsed
    call message('Wall time ray tracing', total_trace_time, 0)
    deallocate(stop_code)
    return

 contains

    integer(c_int32_t) function pick(name, table)
    ! index (0-based) of trim(name) in table; -1 if absent (rays_hip_check_params then rejects it)
       character(len=*), intent(in) :: name
       character(len=*), intent(in) :: table(:)
       integer :: i
       pick = -1
       do i = 1, size(table)
          if (trim(name) == trim(table(i))) pick = i - 1
       end do
    end function pick

 end subroutine trace_rays
