module rays_hip_state_m
! Packs the RAYS host's module state into type(rays_params_t) for librays_hip.so.  Shared by the
! drop-in `trace_rays` (trace_rays_hip.f90) and the drop-in ray launcher
! (solovev_ray_init_hip.f90).  rays_hip_pack_physics fills everything `initialize` has set up by the
! time the ray launcher runs (intialize.f90:50-66: constants, species, rf, damping, equilibrium)
! and gives the ode-solver fields, which are initialised later, neutral values; trace_rays
! overwrites those.  The spline tables of an eqdsk equilibrium are pushed to the library here.

    use, intrinsic :: iso_c_binding
    use rays_hip_m
    implicit none

    ! contiguous TARGET copies of the host's spline objects (c_loc needs a target)
    real(c_double), allocatable, target, private :: t_rg(:), t_zg(:), t_psi(:,:,:,:), t_rbg(:), t_rb(:,:)
    real(c_double), allocatable, target, private :: t_lpsi(:,:), t_lt(:)   ! 'eqdsk_magnetics_lin_interp': Psi, T
    real(c_double), allocatable, target, private :: t_neg(:), t_ne(:,:), t_teg(:), t_te(:,:), t_tig(:), t_ti(:,:)

 contains

    subroutine rays_hip_pack_physics(p, who)
    use constants_m, only : rkind, clight, eps0
    use diagnostics_m, only : integrate_eq_gradients
    use species_m, only : nspec, qs, ms, n0s, t0s, eta
    use rf_m, only : omgrf, k0, ray_param, ray_dispersion_model, dispersion_resid_limit
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
    use solovev_magnetics_m, only : m_rmaj => rmaj, m_kappa => kappa, m_bphi0 => bphi0, m_iota0 => iota0, &
         & m_outer_boundary => outer_boundary, m_psiB => psiB, m_rmin => box_rmin, m_rmax => box_rmax, &
         & m_zmin => box_zmin, m_zmax => box_zmax
    use eqdsk_utilities_m, only : PSIBOUND, NRBOX, NZBOX, eq_R_grid => R_grid, eq_Z_grid => Z_grid, eq_Psi => Psi, &
         & eq_T => T, eq_dR => dR, eq_dZ => dZ
    use density_spline_interp_m, only : ne_profile_N
    use temperature_spline_interp_m, only : Te_profileN, Ti_profileN

    type(rays_params_t), intent(out) :: p
    character(len=*), intent(in) :: who   ! prefix of error messages
    type(rays_axisym_tables_t) :: tab
    character(len=512) :: msg
    integer :: is

    p%abi_version = RAYS_ABI_VERSION
    p%nspec = nspec
    p%pad_ = 0
    p%integrate_eq_gradients = merge(1, 0, integrate_eq_gradients)
    ! ode-solver fields: neutral values until initialize_ode_solver_m has run (trace_rays sets them)
    p%nv = 7 + merge(5, 0, integrate_eq_gradients)
    p%nstep_max = 0
    p%ode_solver = RAYS_ODE_RK4
    p%ray_deriv = RAYS_DERIV_COLD
    p%ds = 0. ; p%s_max = 0.
    p%rel_err0 = 0. ; p%abs_err0 = 0. ; p%SG_error_limit = 0.
    p%damping_model = RAYS_DAMP_NONE
    p%multi_spec_damping = 0
    p%total_damping_limit = 0.

    select case (ray_param)
       case ('arcl'); p%ray_param = RAYS_PARAM_ARCL
       case ('time'); p%ray_param = RAYS_PARAM_TIME
       case default
          write(0,*) 'EQN_RAY: invalid ray parameter = ', ray_param; stop 1
    end select
    if (trim(ray_dispersion_model) /= 'cold') stop 'check_save: unimplemented ray_dispersion_model'

    p%omgrf = omgrf ; p%k0 = k0 ; p%clight = clight ; p%eps0 = eps0
    p%dispersion_resid_limit = dispersion_resid_limit
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
       p%axisym%magnetics_model = pick(magnetics_model, [character(len=32) :: 'eqdsk_magnetics_spline_interp', &
            & 'solovev_magnetics', 'eqdsk_magnetics_lin_interp'])
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
       if (trim(magnetics_model) == 'solovev_magnetics') then
          ! /solovev_magnetics_list/ travels in p%solovev (solovev_magnetics_m.f90:24-35); its own inner box
          ! (the 'R/z out_of_bounds' errors, :147-148) may differ from the one axisym_toroid_eq checks
          p%solovev%rmaj = m_rmaj ; p%solovev%kappa = m_kappa ; p%solovev%bphi0 = m_bphi0
          p%solovev%iota0 = m_iota0 ; p%solovev%outer_bound = m_outer_boundary ; p%solovev%psiB = m_psiB
          p%solovev%box_rmin = m_rmin ; p%solovev%box_rmax = m_rmax
          p%solovev%box_zmin = m_zmin ; p%solovev%box_zmax = m_zmax
          p%axisym%psiB = m_psiB
       end if
       p%axisym%alphan1 = a_alphan1 ; p%axisym%alphan2 = a_alphan2
       p%axisym%d_scrape_off = d_scrape_off ; p%axisym%T_scrape_off = T_scrape_off
       ! spline tables built by initialize_eqdsk_magnetics_spline_interp / initialize_*_spline_interp
       ! (an analytic magnetics model has no psi / R*Bphi tables: nr = nz = n_rb = 0, profile tables only)
       tab%nr = 0 ; tab%nz = 0 ; tab%n_rb = 0
       tab%r_grid = c_null_ptr ; tab%z_grid = c_null_ptr ; tab%psi_fspl = c_null_ptr
       tab%rb_grid = c_null_ptr ; tab%rb_fspl = c_null_ptr
       if (trim(magnetics_model) == 'eqdsk_magnetics_lin_interp') then
          ! eqdsk_utilities_m after initialize_eqdsk_magnetics_lin_interp (Psi already Psi - PSIAXIS, :139)
          t_rg = eq_R_grid ; t_zg = eq_Z_grid ; t_lpsi = eq_Psi ; t_lt = eq_T
          tab%nr = NRBOX ; tab%nz = NZBOX ; tab%n_rb = NRBOX
          tab%r_grid = c_loc(t_rg) ; tab%z_grid = c_loc(t_zg) ; tab%psi_fspl = c_loc(t_lpsi)
          tab%rb_fspl = c_loc(t_lt)
       else if (allocated(Psi_profile%fspl)) then
          t_rg = Psi_profile%x_grid ; t_zg = Psi_profile%y_grid ; t_psi = Psi_profile%fspl
          t_rbg = T_profile%x_grid ; t_rb = T_profile%fspl
          tab%nr = Psi_profile%nx ; tab%nz = Psi_profile%ny ; tab%n_rb = T_profile%nx
          tab%r_grid = c_loc(t_rg) ; tab%z_grid = c_loc(t_zg) ; tab%psi_fspl = c_loc(t_psi)
          tab%rb_grid = c_loc(t_rbg) ; tab%rb_fspl = c_loc(t_rb)
       end if
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
       if (trim(magnetics_model) == 'eqdsk_magnetics_lin_interp') then
          if (rays_hip_set_eqdsk_lin_tables(tab, eq_dR, eq_dZ) /= 0) then
             call last_error_string(msg)
             write(0,*) trim(who)//': ', trim(msg) ; stop 1
          end if
       else if (tab%nr > 0 .or. tab%n_ne > 0 .or. tab%n_te > 0 .or. tab%n_ti > 0) then   ! (analytic everything: no tables)
          if (rays_hip_set_axisym_tables(tab) /= 0) then
             call last_error_string(msg)
             write(0,*) trim(who)//': ', trim(msg) ; stop 1
          end if
       end if
    case default
       write(0,*) trim(who)//': equilib_model not on the device path = ', trim(equilib_model); stop 1
    end select

    return
    end subroutine rays_hip_pack_physics

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

end module rays_hip_state_m
