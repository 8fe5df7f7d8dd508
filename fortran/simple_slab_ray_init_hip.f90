 module simple_slab_ray_init_m
! Drop-in replacement for RAYS_project/RAYS_lib/simple_slab_ray_init_m.f90: same module name, same
! namelist /simple_slab_ray_init_list/, same subroutine
!     simple_slab_ray_init(nray_max, nray, rvec0, rindex_vec0, ray_pwr_wt)
! called from initialize_ray_init_m (ray_init_m.f90:104).  The serial launch loops (:108-164: an
! `equilibrium` call and a dispersion root solve per fan member, evanescent launches dropped) run on the
! GPU through rays_hip_ray_init; rvec0 / rindex_vec0 are bit-identical to the reference's, ray_pwr_wt is
! the reference's 1/nray divided by nray once more (:179-182).
! Compile INSTEAD of the reference file, before ray_init_m.f90 (see INTEGRATION.md).

    use constants_m, only : rkind, one, zero

    implicit none

    integer:: n_x_launch = 1
    real(KIND=rkind) ::  x_launch0 = zero, dx_launch = zero
    integer:: n_y_launch = 1
    real(KIND=rkind) ::  y_launch0 = zero, dy_launch = zero
    integer:: n_z_launch = 1
    real(KIND=rkind) ::  z_launch0 = zero, dz_launch = zero
    integer:: n_ky_launch, n_kz_launch
    real(KIND=rkind) ::  rindex_y0, delta_rindex_y0, rindex_z0, delta_rindex_z0

 namelist /simple_slab_ray_init_list/ &
     & n_x_launch, x_launch0, dx_launch, n_y_launch, y_launch0, dy_launch, &
     & n_z_launch, z_launch0, dz_launch, n_ky_launch, rindex_y0,           &
     & delta_rindex_y0, n_kz_launch, rindex_z0, delta_rindex_z0

contains

    subroutine simple_slab_ray_init(nray_max, nray, rvec0, rindex_vec0, ray_pwr_wt)

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
    read(input_unit, simple_slab_ray_init_list)
    close(unit=input_unit)
    if (verbosity >= 0) then
       write(message_unit, simple_slab_ray_init_list)
       if (messages_to_stdout) write(*, simple_slab_ray_init_list)
    end if

    ! the reference sizes its arrays with n_x * n_ky * n_kz (n_y, n_z are not counted: :108)
    ncand = n_x_launch * n_ky_launch * n_kz_launch
    if (.not. ((ncand > 0) .and. (ncand <= nray_max))) then
       write (*,*) 'simple slab ray init: improper number of rays  nray=', ncand
       stop 1
    end if

    call rays_hip_pack_physics(p, 'simple_slab_ray_init (HIP)')

    call clear_fan(fan)
    fan%model = RAYS_RAY_INIT_SIMPLE_SLAB
    fan%wave_mode = wave_mode_code(wave_mode)
    fan%k0_sign = k0_sign
    fan%n_x_launch = n_x_launch ; fan%n_y_launch = n_y_launch ; fan%n_z_launch = n_z_launch
    fan%n_ky_launch = n_ky_launch ; fan%n_kz_launch = n_kz_launch
    fan%x_launch0 = x_launch0 ; fan%dx_launch = dx_launch
    fan%y_launch0 = y_launch0 ; fan%dy_launch = dy_launch
    fan%slab_z_launch0 = z_launch0          ! z advances by dy_launch in the reference (:119): dz_launch is unused
    fan%rindex_y0 = rindex_y0 ; fan%delta_rindex_y0 = delta_rindex_y0
    fan%rindex_z0 = rindex_z0 ; fan%delta_rindex_z0 = delta_rindex_z0

    allocate(r0(3, nray_max), n0(3, nray_max), w(nray_max))
    if (rays_hip_ray_init(p, fan, int(nray_max, c_int), r0, n0, w, n) /= 0) then
       call last_error_string(msg)
       write(0,*) 'simple_slab_ray_init (HIP): ', trim(msg) ; stop 1
    end if

    nray = n
    allocate ( rvec0(3, nray), rindex_vec0(3, nray) )
    allocate ( ray_pwr_wt(nray) )
    rvec0 = r0(:, 1:nray)
    rindex_vec0 = n0(:, 1:nray)
    ray_pwr_wt = w(1:nray)
    deallocate(r0, n0, w)

    end subroutine simple_slab_ray_init

    subroutine deallocate_simple_slab_ray_init_m
       return ! nothing module-owned to free (as in the reference)
    end subroutine deallocate_simple_slab_ray_init_m

 end module simple_slab_ray_init_m
