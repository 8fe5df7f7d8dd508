 module axisym_toroid_ray_init_R_Z_nphi_ntheta_m
! Drop-in replacement for RAYS_project/RAYS_lib/axisym_toroid_ray_init_R_Z_nphi_ntheta_m.f90: same
! module name, same namelist /axisym_toroid_ray_init_R_Z_nphi_ntheta_list/, same subroutine
!     ray_init_axisym_toroid_R_Z_nphi_ntheta(nray_max, nray, rvec0, rindex_vec0, ray_pwr_wt)
! called from initialize_ray_init_m (ray_init_m.f90:109-111).  The serial launch loops (:139-231:
! `equilibrium` + axisym_toroid_psi + solve_n1_vs_n2_n3 per fan member) run on the GPU through
! rays_hip_ray_init; rvec0 / rindex_vec0 are bit-identical to the reference's, ray_pwr_wt = 1/nray (:241).
! Compile INSTEAD of the reference file, before ray_init_m.f90 (see INTEGRATION.md).

    use constants_m, only : rkind

    implicit none

    integer:: n_R_launch = 1, n_Z_launch = 1
    real(KIND=rkind) ::  R_launch0 = 0., Z_launch0 = 0.
    integer:: n_rindex_theta = 1
    real(KIND=rkind) ::  rindex_theta0 = 0., delta_rindex_theta = 0.
    integer:: n_rindex_phi = 1
    real(KIND=rkind) ::  rindex_phi0 = 0., delta_rindex_phi = 0.

 namelist /axisym_toroid_ray_init_R_Z_nphi_ntheta_list/ &
     & n_R_launch, R_launch0, &
     & n_Z_launch, Z_launch0, &
     & n_rindex_theta, rindex_theta0, delta_rindex_theta, &
     & n_rindex_phi, rindex_phi0, delta_rindex_phi

contains

    subroutine ray_init_axisym_toroid_R_Z_nphi_ntheta(nray_max, nray, rvec0,&
             & rindex_vec0, ray_pwr_wt)

    use, intrinsic :: iso_c_binding
    use diagnostics_m, only: message_unit, messages_to_stdout, verbosity
    use rf_m, only : wave_mode, k0_sign
    use rays_hip_m
    use rays_hip_state_m, only : rays_hip_pack_physics

    implicit none

    integer, intent(in) :: nray_max
    integer, intent(out) :: nray
    real(KIND=rkind), allocatable, intent(out) :: rvec0(:, :), rindex_vec0(:, :)
    real(KIND=rkind), allocatable, intent(out) :: ray_pwr_wt(:)

    integer :: input_unit, get_unit_number ! External, free unit finder
    type(rays_params_t) :: p
    type(rays_fan_t) :: fan
    real(c_double), allocatable :: r0(:,:), n0(:,:), w(:)
    integer(c_int32_t) :: n
    integer :: ncand
    character(len=512) :: msg

    input_unit = get_unit_number()
    open(unit=input_unit, file='rays.in',action='read', status='old', form='formatted')
    read(input_unit, axisym_toroid_ray_init_R_Z_nphi_ntheta_list)
    close(unit=input_unit)
    if (verbosity >= 0) then
       write(message_unit, axisym_toroid_ray_init_R_Z_nphi_ntheta_list)
       if (messages_to_stdout) write(*, axisym_toroid_ray_init_R_Z_nphi_ntheta_list)
    end if

    ncand = n_R_launch * n_Z_launch * n_rindex_theta * n_rindex_phi
    if (.not. ((ncand > 0) .and. (ncand <= nray_max))) then
       write (*,*) 'axisym_toroid ray init: improper number of rays  nray=', ncand
       stop 1
    end if

    ! (pushes the eqdsk spline tables to the library as well)
    call rays_hip_pack_physics(p, 'ray_init_axisym_toroid_R_Z_nphi_ntheta (HIP)')

    call clear_fan(fan)
    fan%model = RAYS_RAY_INIT_AXISYM_R_Z_NPHI_NTHETA
    fan%wave_mode = wave_mode_code(wave_mode)
    fan%k0_sign = k0_sign
    fan%n_r_launch = n_R_launch ; fan%n_theta_launch = n_Z_launch   ! axisym: n_R x n_Z (rays_hip.h: rays_fan_t)
    fan%r_launch0 = R_launch0 ; fan%z_launch0 = Z_launch0
    fan%n_rindex_theta = n_rindex_theta ; fan%n_rindex_phi = n_rindex_phi
    fan%rindex_theta0 = rindex_theta0 ; fan%delta_rindex_theta = delta_rindex_theta
    fan%rindex_phi0 = rindex_phi0 ; fan%delta_rindex_phi = delta_rindex_phi

    allocate(r0(3, ncand), n0(3, ncand), w(ncand))
    if (rays_hip_ray_init(p, fan, int(ncand, c_int), r0, n0, w, n) /= 0) then
       call last_error_string(msg)
       write(0,*) 'ray_init_axisym_toroid_R_Z_nphi_ntheta (HIP): ', trim(msg) ; stop 1
    end if

    nray = n
    allocate ( rvec0(3, nray), rindex_vec0(3, nray) )
    allocate ( ray_pwr_wt(nray) )
    rvec0 = r0(:, 1:nray)
    rindex_vec0 = n0(:, 1:nray)
    ray_pwr_wt = w(1:nray)
    deallocate(r0, n0, w)

    end subroutine ray_init_axisym_toroid_R_Z_nphi_ntheta

    subroutine deallocate_axisym_toroid_ray_init_R_Z_nphi_ntheta_m
       return ! nothing module-owned to free (as in the reference)
    end subroutine deallocate_axisym_toroid_ray_init_R_Z_nphi_ntheta_m

 end module axisym_toroid_ray_init_R_Z_nphi_ntheta_m
