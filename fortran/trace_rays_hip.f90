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
    use constants_m, only : rkind
    use diagnostics_m, only : message, text_message
    use damping_m, only : damping_model, multi_spec_damping, total_damping_limit
    use zfunctions_m, only : fsplRe, zf_nx => nx, x_grid_min, x_grid_max
    use ode_m, only : nv, ds, s_max, nstep_max, ode_solver_name, ray_deriv_name
    use ray_init_m, only : nray, rvec0, rindex_vec0, ray_pwr_wt
    use ray_results_m, only : ray_stop_flag, ray_vec, residual, npoints, end_residuals, &
         & max_residuals, end_ray_parameter, start_ray_vec, end_ray_vec, initial_ray_power, &
         & ray_trace_time, total_trace_time
    use rays_hip_m
    use rays_hip_state_m, only : rays_hip_pack_physics

    implicit none

    type(rays_params_t) :: p
    integer(c_int32_t), allocatable :: stop_code(:)
    integer(c_int) :: rc
    real(c_double) :: elapsed
    character(len=512) :: msg
    integer :: iray
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

    ! species, rf, constants, equilibrium (+ spline tables): shared with the ray launcher shim
    call rays_hip_pack_physics(p, 'trace_rays (HIP)')
    p%nv = nv
    p%nstep_max = nstep_max

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
    p%rel_err0 = rel_err0 ; p%abs_err0 = abs_err0 ; p%SG_error_limit = SG_error_limit

    if (rays_hip_check_params(p) /= 0) then
       call last_error_string(msg)
       write(0,*) 'trace_rays (HIP): ', trim(msg) ; stop 1
    end if

    allocate(stop_code(nray))
    stop_code = 0
    ray_stop_flag = ''

    ! the result's device image stays where it was traced, for a post-processing step of this process
    ! (deposition_profiles_hip.f90: rays_hip_deposition_last) -- RAYS_P.f90:19-44 traces and post-processes in one go
    rc = rays_hip_keep_last_result(1_c_int)

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

 end subroutine trace_rays
