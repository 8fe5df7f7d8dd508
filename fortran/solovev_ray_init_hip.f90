 module solovev_ray_init_nphi_ntheta_m
! Drop-in replacement for RAYS_project/RAYS_lib/solovev_ray_init_nphi_ntheta_m.f90: same module
! name, same namelist /solovev_ray_init_nphi_ktheta_list/, same subroutine
!     ray_init_solovev_nphi_ntheta(nray_max, nray, rvec0, rindex_vec0, ray_pwr_wt)
! called from initialize_ray_init_m (ray_init_m.f90:107).  The serial launch loops
! (two `equilibrium` calls + a dispersion root solve per ray, :123-198) run on the GPU through
! rays_hip_ray_init; the resulting rvec0 / rindex_vec0 are bit-identical to the reference's.
! Compile INSTEAD of the reference file, before ray_init_m.f90 (see INTEGRATION.md).

    use constants_m, only : rkind

    implicit none

    integer:: n_r_launch = 1
    real(KIND=rkind) ::  r_launch0 = 0., dr_launch = 0.
    integer:: n_theta_launch = 1
    real(KIND=rkind) ::  theta_launch0 = 0., dtheta_launch = 0.

    integer:: n_rindex_theta = 1
    real(KIND=rkind) ::  rindex_theta0 = 0., delta_rindex_theta = 0.
    integer:: n_rindex_phi = 1
    real(KIND=rkind) ::  rindex_phi0 = 0., delta_rindex_phi = 0.

 namelist /solovev_ray_init_nphi_ktheta_list/ &
     & n_r_launch,  r_launch0, dr_launch, &
     & n_theta_launch, theta_launch0, dtheta_launch, &
     & n_rindex_theta, rindex_theta0, delta_rindex_theta, &
     & n_rindex_phi, rindex_phi0, delta_rindex_phi

contains

    subroutine ray_init_solovev_nphi_ntheta(nray_max, nray, rvec0, rindex_vec0, ray_pwr_wt)

    use, intrinsic :: iso_c_binding
    use diagnostics_m, only: message_unit
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
    read(input_unit, solovev_ray_init_nphi_ktheta_list)
    close(unit=input_unit)
    write(message_unit, solovev_ray_init_nphi_ktheta_list)

    ncand = n_r_launch * n_theta_launch * n_rindex_theta * n_rindex_phi
    if (.not. ((ncand > 0) .and. (ncand <= nray_max))) then
       write (*,*) 'solovev ray init: improper number of rays  nray=', ncand
       stop 1
    end if

    call rays_hip_pack_physics(p, 'ray_init_solovev_nphi_ntheta (HIP)')

    fan%model = RAYS_RAY_INIT_SOLOVEV_NPHI_NTHETA
    select case (trim(wave_mode))
       case ('plus');  fan%wave_mode = RAYS_WAVE_PLUS
       case ('minus'); fan%wave_mode = RAYS_WAVE_MINUS
       case ('fast');  fan%wave_mode = RAYS_WAVE_FAST
       case ('slow');  fan%wave_mode = RAYS_WAVE_SLOW
       case default
          write(0,*) 'solve_disp: improper wave_mode = ', trim(wave_mode); stop 1
    end select
    fan%k0_sign = k0_sign
    fan%n_r_launch = n_r_launch ; fan%n_theta_launch = n_theta_launch
    fan%n_rindex_theta = n_rindex_theta ; fan%n_rindex_phi = n_rindex_phi
    fan%r_launch0 = r_launch0 ; fan%dr_launch = dr_launch
    fan%theta_launch0 = theta_launch0 ; fan%dtheta_launch = dtheta_launch ; fan%z_launch0 = 0.
    fan%rindex_theta0 = rindex_theta0 ; fan%delta_rindex_theta = delta_rindex_theta
    fan%rindex_phi0 = rindex_phi0 ; fan%delta_rindex_phi = delta_rindex_phi
    fan%n_x_launch = 0 ; fan%n_y_launch = 0 ; fan%n_z_launch = 0 ; fan%n_ky_launch = 0
    fan%n_kz_launch = 0 ; fan%pad_ = 0
    fan%x_launch0 = 0. ; fan%dx_launch = 0. ; fan%y_launch0 = 0. ; fan%dy_launch = 0.
    fan%slab_z_launch0 = 0. ; fan%rindex_y0 = 0. ; fan%delta_rindex_y0 = 0.
    fan%rindex_z0 = 0. ; fan%delta_rindex_z0 = 0.

    allocate(r0(3, ncand), n0(3, ncand), w(ncand))
    if (rays_hip_ray_init(p, fan, int(ncand, c_int), r0, n0, w, n) /= 0) then
       call last_error_string(msg)
       write(0,*) 'ray_init_solovev_nphi_ntheta (HIP): ', trim(msg) ; stop 1
    end if

    ! the reference allocates for the candidate count and sets nray to the survivors (:111-117, 198)
    nray = n
    allocate ( rvec0(3, ncand), rindex_vec0(3, ncand) )
    allocate ( ray_pwr_wt(ncand) )
    rvec0 = 0. ; rindex_vec0 = 0. ; ray_pwr_wt = 0.
    rvec0(:, 1:nray) = r0(:, 1:nray)
    rindex_vec0(:, 1:nray) = n0(:, 1:nray)
    ray_pwr_wt(1:nray) = w(1:nray)
    deallocate(r0, n0, w)

    end subroutine ray_init_solovev_nphi_ntheta

    subroutine deallocate_solovev_ray_init_nphi_ntheta_m
    ! nothing module-owned to free (as in the reference, :285-292)
       return
    end subroutine deallocate_solovev_ray_init_nphi_ntheta_m

 end module solovev_ray_init_nphi_ntheta_m
