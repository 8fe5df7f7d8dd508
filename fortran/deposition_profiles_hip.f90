 module deposition_profiles_hip_m
! GPU form of the reference post-processor's profile binning
!     calculate_deposition_profiles / bin_a_ray      post_process_lib/deposition_profiles_m.f90:228-292
! for a host that holds the ray_results_m arrays (post_process_RAYS after read_results, or RAYS itself right
! after trace_rays).  One call per profile:
!     call deposition_profile_hip(profile_name, n_bins, work, profile, Q_sum)
! fills work(n_bins, nray), profile(n_bins) = sum(work, 2) in ray order and Q_sum = sum(profile) exactly as the
! reference does (bit-identical: tests/test_gpu_parity.py::test_device_deposition_profiles_match_reference), for
!     'Ptotal_psi', 'Ptotal_rho'   equilib_model = 'axisym_toroid'   (:170-222; grid [0, 1])
!     'Ptotal_x'                   equilib_model = 'slab'            (:134-160; grid [xmin, xmax])
! The rho(psiN) spline of 'Ptotal_rho' must have been handed over with rays_hip_set_rho_table (the host's
! eqdsk_magnetics_spline_interp_m: rho_profile).

    use, intrinsic :: iso_c_binding
    use rays_hip_m

    implicit none

 contains

    subroutine deposition_profile_hip(profile_name, n_bins, work, profile, Q_sum)

    use constants_m, only : rkind
    use ray_results_m, only : number_of_rays, ray_vec, npoints, initial_ray_power
    use rays_hip_state_m, only : rays_hip_pack_physics

    character(len=*), intent(in) :: profile_name
    integer, intent(in) :: n_bins
    real(KIND=rkind), intent(out) :: work(n_bins, number_of_rays), profile(n_bins), Q_sum

    type(rays_params_t) :: p
    integer(c_int) :: which
    integer(c_int32_t), allocatable :: np32(:)
    character(len=512) :: msg
    integer :: i, m, g
    integer(c_int) :: rc
    logical :: found

    select case (trim(profile_name))
       case ('Ptotal_psi'); which = RAYS_DEP_PTOTAL_PSI
       case ('Ptotal_rho'); which = RAYS_DEP_PTOTAL_RHO
       case ('Ptotal_x');   which = RAYS_DEP_PTOTAL_X
       case default
          write(0,*) 'initialize_deposition_profiles: unimplemented profile ', trim(profile_name) ; stop 1
    end select

    call rays_hip_pack_physics(p, 'deposition_profile_hip')
    ! the run's ODE vector length and array extent, from the arrays themselves (a post-processor has read them
    ! from the results file): v(8) = absorbed power fraction needs a run with damping (ode_m.f90:160-173)
    p%nv = size(ray_vec, 1)
    p%nstep_max = size(ray_vec, 2) - 1
    ! The binning reads v(1:3) and v(8) only; nv is the stride.  Any run with damping has the v(8) row:
    ! nv = 8 [+ 1 + nspec with multi_spec_damping] [+ 5 with integrate_eq_gradients] (ode_m.f90:160-173).  A post-
    ! processor has only the arrays, so the options are recovered from nv (where two combinations give the same nv
    ! -- 13 = 8 + 5 = 8 + 1 + 4 species -- either describes the same rows 1..8).
    if (p%nv < 8) then
       write(0,*) 'deposition_profile_hip: ray_vec carries no absorbed-power row (dim_v_vector =', p%nv, ')' ; stop 1
    end if
    p%damping_model = RAYS_DAMP_FUND_ECH
    found = .false.
    do m = 0, 1
       do g = 0, 1
          if (.not. found .and. 8 + m*(1 + p%nspec) + 5*g == p%nv) then
             p%multi_spec_damping = m
             p%integrate_eq_gradients = g
             found = .true.
          end if
       end do
    end do
    if (.not. found) then
       write(0,*) 'deposition_profile_hip: dim_v_vector =', p%nv, ' matches no combination of damping options for nspec =', p%nspec
       stop 1
    end if

    ! If this process traced these rays itself (fortran/trace_rays_hip.f90), their trajectories still lie on the GPU(s)
    ! that traced them: binned there, nothing is uploaded.  Otherwise (a post-processor that read the arrays from a
    ! results file) the arrays go to the device: only points 1..maxval(npoints) of each ray cross PCIe.
    rc = rays_hip_deposition_last(p, which, int(n_bins, c_int), int(number_of_rays, c_int), initial_ray_power, work, profile)
    if (rc == RAYS_HIP_NO_KEPT_RESULT) then
       allocate(np32(number_of_rays))
       np32 = npoints
       rc = rays_hip_deposition(p, which, int(n_bins, c_int), int(number_of_rays, c_int), ray_vec, np32, &
            & initial_ray_power, work, profile)
       deallocate(np32)
    end if
    if (rc /= 0) then
       call last_error_string(msg)
       write(0,*) 'deposition_profile_hip: ', trim(msg) ; stop 1
    end if

    Q_sum = 0.
    do i = 1, n_bins          ! sum(profile), in order
       Q_sum = Q_sum + profile(i)
    end do

    end subroutine deposition_profile_hip

 end module deposition_profiles_hip_m
