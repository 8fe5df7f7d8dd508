! TEST INFRASTRUCTURE ONLY.
! Driver written for this repo (not reference code): links against the reference RAYS_lib
! objects built by oracle/build_ref.sh, runs the reference's own
!     initialize -> trace_rays      (RAYS_project/RAYS_code/RAYS.f90:10-16)
! and dumps module state + ray_results_m arrays as one raw little-endian stream file so that
! golden vectors (tests/golden/) can be cut from it and so the CPU reference path can be timed
! (omp_get_wtime around trace_rays only; SURVEY.md 8(d)).
!
! Controlled by environment variables (the reference allows at most one argv = namelist file):
!   RAYS_DUMP_FILE   output path (default ref_dump.bin); 'none' = no dump, timing only;
!                    'skip' = no trajectory dump, but the other dumps below (deposition profiles ...) are made
!   RAYS_DUMP_PROBE  stride (in recorded points) for per-state probes of equilibrium,
!                    deriv_cold, deriv_num, eqn_ray, check_save; 0/unset = no probes
!   RAYS_DUMP_REPS   repeat trace_rays this many times for timing (default 1)
!   RAYS_DUMP_DEPOSITION  output path for the deposition profiles of the run (reference
!                    post_process_lib/deposition_profiles_m applied to the ray_results_m arrays;
!                    slab and axisym_toroid only): per-ray binned arrays + summed profiles
!
! When linked as oracle/_ref/rays_hip_dropin the same driver runs with trace_rays replaced by
! fortran/trace_rays_hip.f90 (the C-ABI drop-in), so both binaries write the same format.
program ref_dump_driver
    use damping_m, only : damping_model
    use solovev_magnetics_m, only : m_psiB => psiB
    use constants_m, only : rkind, clight, eps0
    use species_m, only : nspec, qs, ms, n0s, t0s, eta
    use rf_m, only : omgrf, k0, dispersion_resid_limit
    use ode_m, only : nv, ds, s_max, nstep_max, ode_stop
    use ray_init_m, only : nray, rvec0, rindex_vec0
    use ray_results_m, only : ray_vec, residual, npoints, ray_stop_flag, end_ray_vec, initial_ray_power, &
         & write_results_LD
    use deposition_profiles_m, only : write_deposition_profiles_LD, initialize_deposition_profiles, calculate_deposition_profiles, &
         & bin_a_ray, profiles_1D, n_profiles
    use equilibrium_m, only : equilibrium, eq_point, equilib_model
    use solovev_eq_m, only : rmaj, kappa, bphi0, iota0, outer_bound, psiB
    use zfunctions_m, only : fsplRe, zf_nx => nx, x_grid_min, x_grid_max
    use axisym_toroid_eq_m, only : ax_rmin => box_rmin, ax_rmax => box_rmax, ax_zmin => box_zmin, &
         & ax_zmax => box_zmax, plasma_psi_limit, magnetics_model
    use eqdsk_magnetics_spline_interp_m, only : Psi_profile, T_profile, rho_profile
    use eqdsk_utilities_m, only : PSIBOUND, NRBOX, NZBOX, eq_dR => dR, eq_dZ => dZ, eq_R_grid => R_grid, &
         & eq_Z_grid => Z_grid, eq_Psi => Psi, eq_T => T
    use density_spline_interp_m, only : ne_profile_N
    use temperature_spline_interp_m, only : Te_profileN, Ti_profileN
    use omp_lib
#ifdef RAYS_DROPIN
    use, intrinsic :: iso_c_binding
    use rays_hip_m, only : rays_hip_set_rho_table
    use deposition_profiles_hip_m, only : deposition_profile_hip
#endif
    implicit none

    logical :: read_input = .true.
    character(len=256) :: fname, sval
    integer :: u, u2, stat, probe_stride, reps, irep, iray, j, nprobe, is, n_ne, n_te, n_ti, ip, np_dump
    real(kind=rkind) :: t0, t1, wall, resid, s
    real(kind=rkind), allocatable :: v(:), dvds(:)
    real(kind=rkind) :: dddx(3), dddk(3), dddw, ndx(3), ndk(3), ndw, nvec(3)
    type(eq_point) :: eq
    type(ode_stop) :: rs
    integer(kind=8) :: total_steps
#ifdef RAYS_DROPIN
    real(kind=rkind), allocatable :: hwork(:,:), hprofile(:,:), hq(:)
#endif

    interface
       subroutine deriv_cold(eq, nvec, dddx, dddk, dddw)
          use constants_m, only : rkind
          use equilibrium_m, only : eq_point
          type(eq_point), intent(in) :: eq
          real(KIND=rkind), intent(in) :: nvec(3)
          real(KIND=rkind), intent(out) :: dddx(3), dddk(3), dddw
       end subroutine deriv_cold
       subroutine deriv_num(eq, v, dddx, dddk, dddw)
          use constants_m, only : rkind
          use equilibrium_m, only : eq_point
          use ode_m, only : nv
          type(eq_point), intent(in) :: eq
          real(KIND=rkind), intent(in) :: v(nv)
          real(KIND=rkind), intent(out) :: dddx(3), dddk(3), dddw
       end subroutine deriv_num
       subroutine eqn_ray(s, v, dvds, ray_stop)
          use constants_m, only : rkind
          use ode_m, only : nv, ode_stop
          real(KIND=rkind), intent(in) :: s
          real(KIND=rkind), intent(in) :: v(nv)
          real(KIND=rkind), intent(out) :: dvds(nv)
          type(ode_stop) :: ray_stop
       end subroutine eqn_ray
       subroutine check_save(s, nv, v, resid, ray_stop)
          use constants_m, only : rkind
          use ode_m, only : ode_stop
          real(KIND=rkind), intent(in) :: s
          integer, intent(in) :: nv
          real(KIND=rkind), intent(in) :: v(nv)
          real(KIND=rkind), intent(out) :: resid
          type(ode_stop) :: ray_stop
       end subroutine check_save
    end interface

    fname = 'ref_dump.bin'
    call get_environment_variable('RAYS_DUMP_FILE', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0) fname = sval
    probe_stride = 0
    call get_environment_variable('RAYS_DUMP_PROBE', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0) read(sval, *) probe_stride
    reps = 1
    call get_environment_variable('RAYS_DUMP_REPS', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0) read(sval, *) reps

    call initialize(read_input)

    wall = huge(wall)
    do irep = 1, reps
       if (irep > 1) then      ! results arrays are expected zero-filled (ray_results_m.f90:154-164)
          ray_vec = 0. ; residual = 0. ; npoints = 0 ; end_ray_vec = 0. ; ray_stop_flag = ''
       end if
       t0 = omp_get_wtime()
       call trace_rays
       t1 = omp_get_wtime()
       wall = min(wall, t1 - t0)
    end do

    total_steps = 0
    do iray = 1, nray
       total_steps = total_steps + max(npoints(iray) - 1, 0)
    end do
    write(*,'(a,i0)')     'RAYS_REF nray = ', nray
    write(*,'(a,i0)')     'RAYS_REF total_steps = ', total_steps
    write(*,'(a,es16.8)') 'RAYS_REF trace_wall_s = ', wall
    write(*,'(a,i0)')     'RAYS_REF threads = ', omp_get_max_threads()
    write(*,'(a,es16.8)') 'RAYS_REF steps_per_s = ', real(total_steps, rkind)/wall

    call get_environment_variable('RAYS_DUMP_RESULTS_LD', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0) call write_results_LD   ! the reference's own run_results.<label>

    if (trim(fname) == 'none') stop

    if (trim(fname) /= 'skip') then
    open(newunit=u, file=trim(fname), access='stream', form='unformatted', status='replace')
    write(u) int(z'52415953'), 2, nray, nv, nstep_max, nspec, probe_stride, 0
    write(u) omgrf, k0, clight, eps0, ds, s_max, dispersion_resid_limit, wall
    write(u) qs(0:5), ms(0:5), n0s(0:5), t0s(0:5), eta(0:5)
    write(u) rmaj, kappa, bphi0, iota0, outer_bound, psiB
    write(u) rvec0(1:3,1:nray), rindex_vec0(1:3,1:nray)
    write(u) npoints(1:nray)
    write(u) ray_stop_flag(1:nray)
    write(u) ray_vec
    write(u) residual
    write(u) end_ray_vec
    end if

    call get_environment_variable('RAYS_DUMP_ZFUN', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0) then   ! Z-function spline table (zfunctions_m)
       open(newunit=u2, file=trim(sval), access='stream', form='unformatted', status='replace')
       write(u2) zf_nx
       write(u2) x_grid_min, x_grid_max
       write(u2) fsplRe
       close(u2)
    end if

    call get_environment_variable('RAYS_DUMP_DEPOSITION', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0 .and. nv >= 8 .and. trim(damping_model) /= 'no_damp' .and. &
      & (trim(equilib_model) == 'axisym_toroid' .or. trim(equilib_model) == 'slab')) then
       call initialize_deposition_profiles(.false.)     ! default n_bins (no post_process_rays.in)
       ! axisym_toroid with an analytic magnetics model has no rho(psiN) spline and the reference's
       ! axisym_toroid_rho does not implement it (axisym_toroid_eq_m.f90:398-430): 'Ptotal_psi' (profile 1) only
       np_dump = n_profiles
       if (trim(equilib_model) == 'axisym_toroid' .and. .not. allocated(rho_profile%fspl)) np_dump = 1
       open(newunit=u2, file=trim(sval), access='stream', form='unformatted', status='replace')
       write(u2) np_dump, profiles_1D(1)%n_bins, nray
       write(u2) initial_ray_power(1:nray)
       if (trim(equilib_model) == 'axisym_toroid' .and. allocated(rho_profile%fspl)) then   ! rho(psiN) spline of the eqdsk equilibrium
          write(u2) rho_profile%nx
          write(u2) rho_profile%x_grid, rho_profile%fspl
       else
          write(u2) 0
       end if
#ifdef RAYS_DROPIN
       ! drop-in binary: the profiles come from the GPU (fortran/deposition_profiles_hip.f90), same record layout
       if (trim(equilib_model) == 'axisym_toroid' .and. allocated(rho_profile%fspl)) then
          if (rays_hip_set_rho_table(rho_profile%x_grid, rho_profile%fspl, int(rho_profile%nx, c_int)) /= 0) stop 1
       end if
       allocate(hwork(profiles_1D(1)%n_bins, nray), hprofile(profiles_1D(1)%n_bins, n_profiles), hq(n_profiles))
       do ip = 1, np_dump
          call deposition_profile_hip(trim(profiles_1D(ip)%profile_name), profiles_1D(ip)%n_bins, hwork, &
               & hprofile(:, ip), hq(ip))
          write(u2) profiles_1D(ip)%profile_name, profiles_1D(ip)%grid_min, profiles_1D(ip)%grid_max
          write(u2) hwork
       end do
       do ip = 1, np_dump
          write(u2) hprofile(:, ip), hq(ip)
       end do
       close(u2)
#else
       do ip = 1, np_dump             ! per-ray binned arrays, as calculate_deposition_profiles fills them
          do iray = 1, nray
             call bin_a_ray(profiles_1D(ip), iray)
          end do
          write(u2) profiles_1D(ip)%profile_name, profiles_1D(ip)%grid_min, profiles_1D(ip)%grid_max
          write(u2) profiles_1D(ip)%work
       end do
       if (np_dump == n_profiles) then
          call calculate_deposition_profiles             ! the reference's own sums
       else                                              ! its two statements for the profiles that exist (:250-251)
          do ip = 1, np_dump
             profiles_1D(ip)%profile(:) = sum(profiles_1D(ip)%work, 2)
             profiles_1D(ip)%Q_sum = sum(profiles_1D(ip)%profile)
          end do
       end if
       do ip = 1, np_dump
          write(u2) profiles_1D(ip)%profile, profiles_1D(ip)%Q_sum
       end do
       close(u2)
#endif
       ! RAYS_DUMP_DEPOSITION_LD: the reference's own list-directed profile file, deposition_profiles.<run_label>
       ! (write_deposition_profiles_LD, deposition_profiles_m.f90:296-331) -- a data fixture for the writer of
       ! rays_amd/results.py.  In the drop-in binary the profiles it prints are the GPU's.
       call get_environment_variable('RAYS_DUMP_DEPOSITION_LD', sval, status=stat)
       if (stat == 0 .and. len_trim(sval) > 0 .and. np_dump == n_profiles) then
#ifdef RAYS_DROPIN
          do ip = 1, n_profiles
             profiles_1D(ip)%profile = hprofile(:, ip)
             profiles_1D(ip)%Q_sum = hq(ip)
          end do
#endif
          call write_deposition_profiles_LD
       end if
    end if

    call get_environment_variable('RAYS_DUMP_AXISYM', sval, status=stat)
    if (stat == 0 .and. len_trim(sval) > 0 .and. trim(equilib_model) == 'axisym_toroid') then
       ! spline tables of the eqdsk equilibrium (host objects of quick_cube_splines_m); an analytic magnetics
       ! model has only the profile tables (nr = nz = n_rb = 0)
       open(newunit=u2, file=trim(sval), access='stream', form='unformatted', status='replace')
       n_ne = 0 ; n_te = 0 ; n_ti = 0
       if (allocated(ne_profile_N%fspl)) n_ne = ne_profile_N%nx
       if (allocated(Te_profileN%fspl)) n_te = Te_profileN%nx
       if (allocated(Ti_profileN%fspl)) n_ti = Ti_profileN%nx
       if (trim(magnetics_model) == 'eqdsk_magnetics_lin_interp') then
          ! eqdsk_utilities_m after initialize_eqdsk_magnetics_lin_interp (n_rb = -1 marks this layout)
          write(u2) NRBOX, NZBOX, -1, n_ne, n_te, n_ti
          write(u2) ax_rmin, ax_rmax, ax_zmin, ax_zmax, plasma_psi_limit, PSIBOUND
          write(u2) eq_dR, eq_dZ
          write(u2) eq_R_grid, eq_Z_grid, eq_Psi
          write(u2) eq_T
       else if (allocated(Psi_profile%fspl)) then
          write(u2) Psi_profile%nx, Psi_profile%ny, T_profile%nx, n_ne, n_te, n_ti
          write(u2) ax_rmin, ax_rmax, ax_zmin, ax_zmax, plasma_psi_limit, PSIBOUND
          write(u2) Psi_profile%x_grid, Psi_profile%y_grid, Psi_profile%fspl
          write(u2) T_profile%x_grid, T_profile%fspl
       else
          write(u2) 0, 0, 0, n_ne, n_te, n_ti
          write(u2) ax_rmin, ax_rmax, ax_zmin, ax_zmax, plasma_psi_limit, m_psiB   ! psiB of solovev_magnetics_m
       end if
       if (n_ne > 0) write(u2) ne_profile_N%x_grid, ne_profile_N%fspl
       if (n_te > 0) write(u2) Te_profileN%x_grid, Te_profileN%fspl
       if (n_ti > 0) write(u2) Ti_profileN%x_grid, Ti_profileN%fspl
       close(u2)
    end if

    ! ---- optional per-state probes of the RHS pieces (unit parity for the restatement) ----
    if (probe_stride > 0 .and. trim(fname) /= 'skip') then
       allocate(v(nv), dvds(nv))
       nprobe = 0
       do iray = 1, nray
          do j = 1, npoints(iray), probe_stride
             nprobe = nprobe + 1
          end do
       end do
       write(u) nprobe
       do iray = 1, nray
          do j = 1, npoints(iray), probe_stride
             v = ray_vec(:, j, iray)
             s = (j-1)*ds
             call equilibrium(v(1:3), eq)
             nvec = v(4:6)/k0
             call deriv_cold(eq, nvec, dddx, dddk, dddw)
             call deriv_num(eq, v, ndx, ndk, ndw)
             rs%stop_ode = .false. ; rs%ode_stop_flag = ''
             dvds = 0.
             call eqn_ray(s, v, dvds, rs)
             rs%stop_ode = .false. ; rs%ode_stop_flag = ''
             call check_save(s, nv, v, resid, rs)
             write(u) iray, j
             write(u) v
             write(u) eq%bvec, eq%bmag, eq%gradbmag, eq%bunit, eq%gradbunit, eq%gradbtensor
             do is = 0, nspec
                write(u) eq%ns(is), eq%gradns(:,is), eq%ts(is), eq%gradts(:,is), &
                       & eq%omgc(is), eq%omgp2(is), eq%alpha(is), eq%gamma(is)
             end do
             write(u) dddx, dddk, dddw
             write(u) ndx, ndk, ndw
             write(u) dvds
             write(u) resid
          end do
       end do
    end if
    if (trim(fname) /= 'skip') close(u)
end program ref_dump_driver
