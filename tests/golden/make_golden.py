#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE binary.

Runs oracle/_ref/rays_ref_dump (the reference RAYS_project hot path compiled from
/root/reference by oracle/build_ref.sh, amdflang -O2 -ffp-contract=off) on the namelists in
configs/, and cuts small fixtures (inputs + expected outputs, data only) from its raw dump.
Only runs where /root/reference was available to build the binary; the committed .npz files are
what the tests read.

    python tests/golden/make_golden.py [fixture names ...]    (re)generate
    python tests/golden/make_golden.py --check                 regenerate everything into a scratch directory and
                                                               compare it with the committed files bit for bit

The host-side spline tables next to an eqdsk file (configs/<file>.geqdsk.tables.npz, read by the Python host) are the
UNION of what the cases on that file dump (each case's namelist selects its own profile splines: ne_* in one,
te_* / ti_* in another); overlapping entries must agree bit for bit.  Two files: solovev_65x65.geqdsk (written by
tools/make_solovev_eqdsk.py) and solovev_129x129.geqdsk (BASELINE config 5's, written by the REFERENCE's own
solovev_2_eqdsk: oracle/make_cfg5_eqdsk.sh).
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.refdump import read_axisym_tables, read_deposition, read_dump  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "rays_ref_dump")

# (fixture name, config, ray subset (None = all), probe stride, n probes kept)
CASES = [
    ("cfg1_slab16_rk4", "cfg1_slab16_rk4.in", None, 25, 200),
    ("cfg2_solovev1024_rk4", "cfg2_solovev1024_rk4.in", list(range(0, 1024, 33)) + [1023], 40, 400),
    ("gold_solovev64_sg_cold", "gold_solovev64_sg_cold.in", list(range(0, 64, 5)), 0, 0),
    ("gold_solovev64_sg_num", "gold_solovev64_sg_num.in", list(range(0, 64, 5)), 0, 0),
    ("gold_solovev64_rk4_num", "gold_solovev64_rk4_num.in", list(range(0, 64, 5)), 0, 0),
    # the other root of the launcher's dispersion solve: wave_mode = 'slow' (Solovev, SG) and 'fast' (slab, RK4)
    ("gold_solovev64_slow_sg", "gold_solovev64_slow_sg.in", list(range(0, 64, 7)), 0, 0),
    ("gold_slab16_fast_rk4", "gold_slab16_fast_rk4.in", None, 0, 0),
    # slab with a fundamental ECH resonance layer: damping on the slab + the 'Ptotal_x' deposition profile
    ("gold_slab16_damp_rk4", "gold_slab16_damp_rk4.in", None, 0, 0),
    # stop flags the other fixtures do not reach: 'y out_of_bounds', 'z out_of_bounds', 'negative_temp', 'negative_dens'
    ("gold_slab_box_exits_rk4", "gold_slab_box_exits_rk4.in", None, 0, 0),
    ("gold_slab_negative_temp_rk4", "gold_slab_negative_temp_rk4.in", None, 0, 0),
    ("gold_slab_negative_dens_rk4", "gold_slab_negative_dens_rk4.in", None, 0, 0),
    # ray_param = 'arcl' + integrate_eq_gradients: the nv = 12 SG kernel on the Solovev equilibrium
    ("gold_solovev64_arcl_grad_sg", "gold_solovev64_arcl_grad_sg.in", list(range(0, 64, 9)), 0, 0),
    # non-unit profile exponents: the general (libm pow) kernels
    ("gold_solovev64_pow_rk4", "gold_solovev64_pow_rk4.in", list(range(0, 64, 5)), 20, 120),
    # a fan reaching past the cutoff: evanescent launches dropped by the ray launcher (ray numbering)
    ("gold_solovev_evanescent_rk4", "gold_solovev_evanescent_rk4.in", None, 0, 0),
    # fundamental-ECH damping (damp_fund_ECH, nv = 8): constant density + parabolic Te, B0 = 3.3 T
    ("gold_solovev64_damp_rk4", "gold_solovev64_damp_rk4.in", list(range(0, 64, 5)), 0, 0),
    ("gold_solovev64_damp_sg", "gold_solovev64_damp_sg.in", list(range(0, 64, 5)), 0, 0),
    # axisym_toroid: eqdsk bicubic-spline magnetics (configs/solovev_65x65.geqdsk, written by
    # tools/make_solovev_eqdsk.py) + splined density + parabolic Te + ECH damping; the fixture also
    # carries the host-built spline tables (RAYS_DUMP_AXISYM)
    ("gold_axisym64_eqdsk_damp_rk4", "gold_axisym64_eqdsk_damp_rk4.in", list(range(0, 64, 5)), 10, 120),
    # the same with the Shampine-Gordon integrator: BASELINE config 5's kernel
    ("gold_axisym64_eqdsk_damp_sg", "gold_axisym64_eqdsk_damp_sg.in", list(range(0, 64, 5)), 0, 0),
    # steep poloidal launch: every ray ends on 'out_of_plasma'
    ("gold_axisym16_eqdsk_zexit_rk4", "gold_axisym16_eqdsk_zexit_rk4.in", None, 0, 0),
    # parabolic density with non-unit exponents (pow), splined Te and Ti, finite-difference dD
    ("gold_axisym64_eqdsk_tspline_rk4_num", "gold_axisym64_eqdsk_tspline_rk4_num.in", list(range(0, 64, 5)), 10, 60),
    # the slab models cfg 1 does not touch: toroid By/Bz + parabolic n (libm pow) and Te, ray_param = 'arcl',
    # integrate_eq_gradients (nv = 12) | sheared By + linear_2 Bz + Gaussian n + two ion species, SG with
    # finite-difference dD | linear_2 n and Te (whose value and gradient disagree in the reference), RK4 numerical
    ("gold_slab_toroid_parab_arcl_grad_rk4", "gold_slab_toroid_parab_arcl_grad_rk4.in", None, 30, 120),
    ("gold_slab_shear_gauss_3spec_sg_num", "gold_slab_shear_gauss_3spec_sg_num.in", None, 0, 0),
    ("gold_slab_lin2_rk4_num", "gold_slab_lin2_rk4_num.in", None, 25, 60),
    # kernel shapes no other fixture instantiates: electrons only (nspec = 0 -> NS = 1; the reference needs
    # `neutrality` relaxed for that) and damping + integrate_eq_gradients (nv = 13)
    ("gold_slab_ns1_rk4", "gold_slab_ns1_rk4.in", None, 25, 60),
    ("gold_solovev64_damp_grad_rk4", "gold_solovev64_damp_grad_rk4.in", list(range(0, 64, 5)), 0, 0),
    # more ion species: D + H + T + 3He + alpha (nspec = 5 -> NS = 6, SG) and D + T + 3He (NS = 4, RK4 numerical)
    ("gold_slab_6spec_sg", "gold_slab_6spec_sg.in", None, 0, 0),
    ("gold_solovev64_4spec_rk4_num", "gold_solovev64_4spec_rk4_num.in", list(range(0, 64, 5)), 0, 0),
    # multi_spec_damping: one absorbed-power row per species behind the total (nv = 10 | 15; ode_m.f90:169)
    ("gold_solovev64_damp_multi_sg", "gold_solovev64_damp_multi_sg.in", list(range(0, 64, 5)), 0, 0),
    ("gold_slab16_damp_multi_grad_rk4", "gold_slab16_damp_multi_grad_rk4.in", None, 0, 0),
    # axisym_toroid with the analytic Solovev field as its magnetics model (solovev_magnetics_m.f90)
    ("gold_axisym64_solmag_damp_rk4", "gold_axisym64_solmag_damp_rk4.in", list(range(0, 64, 5)), 10, 60),
    ("gold_axisym64_solmag_sg_num", "gold_axisym64_solmag_sg_num.in", list(range(0, 64, 7)), 0, 0),
    ("gold_axisym64_solmag_splines_grad_rk4", "gold_axisym64_solmag_splines_grad_rk4.in", list(range(0, 64, 9)), 25, 0),
    # BASELINE config 5 / 5b as SURVEY 8(d) wrote it, 8 x 8 rays: the 129 x 129 g-eqdsk written by the reference's own
    # solovev_2_eqdsk (oracle/make_cfg5_eqdsk.sh), splined n, Te and Ti, ECH damping; RK4 and Shampine-Gordon
    ("gold_axisym64_eqdsk129_tspline_damp_rk4", "gold_axisym64_eqdsk129_tspline_damp_rk4.in", list(range(0, 64, 5)), 10, 60),
    ("gold_axisym64_eqdsk129_tspline_damp_sg", "gold_axisym64_eqdsk129_tspline_damp_sg.in", list(range(0, 64, 5)), 0, 0),
    ("gold_axisym64_eqlin_damp_rk4", "gold_axisym64_eqlin_damp_rk4.in", list(range(0, 64, 5)), 10, 0),
    ("gold_axisym64_eqlin_tspline_sg_num", "gold_axisym64_eqlin_tspline_sg_num.in", list(range(0, 64, 7)), 0, 0),
    # launchers that take single rays by position and direction (host-side in every build: acos / cos of libm)
    ("gold_slab_one_ray_rk4", "gold_slab_one_ray_rk4.in", None, 0, 0),
    ("gold_solovev_file_rays_damp_rk4", "gold_solovev_file_rays_damp_rk4.in", None, 0, 0),
]


def check_deterministic(cfg):
    """The reference's Z-function spline keeps SAVEd work arrays (zfunctions_m.f90:378,393) and
    deriv_num rewrites module variables (deriv_num.f90:6-10): under the OpenMP ray loop both race and the
    output varies from run to run.  A fixture of such a config must be cut from a one-thread run."""
    from rays_amd.namelist import read_namelist
    nml = read_namelist(os.path.join(ROOT, "configs", cfg))
    racy = str(nml.get("damping_list", {}).get("damping_model", "no_damp")).strip() != "no_damp" or \
        str(nml.get("ode_list", {}).get("ray_deriv_name", "cold")).strip() == "numerical"
    if racy and int(nml.get("openmp_list", {}).get("num_threads", 0)) != 1:
        sys.exit(f"configs/{cfg}: damping / numerical-derivative configs need `&openmp_list num_threads = 1 /` "
                 "(the reference races otherwise and the fixture would not be reproducible)")


EQDSK_FILES = ("solovev_65x65.geqdsk", "solovev_129x129.geqdsk")


def tables_path(eqdsk):
    return os.path.join("configs", eqdsk + ".tables.npz")


def eqdsk_of(cfg):
    """the g-eqdsk file a namelist's spline magnetics read (None: another magnetics model)"""
    from rays_amd.namelist import read_namelist
    nml = read_namelist(os.path.join(ROOT, "configs", cfg))
    f = str(nml.get("eqdsk_magnetics_spline_interp_list", {}).get("eqdsk_file_name", "")).strip()
    return f or None


LD_FIXTURE = os.path.join("tests", "golden", "run_results.gold_slab4_ld")
DEP_LD_FIXTURE = os.path.join("tests", "golden", "deposition_profiles.gaxi")


def results_file_fixture(out_root):
    """tests/golden/run_results.gold_slab4_ld: the reference's own write_results_LD output
    (ray_results_m.f90:365-420) for configs/gold_slab4_results_ld.in -- data for the writers of
    rays_amd/results.py."""
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(os.path.join(ROOT, "configs", "gold_slab4_results_ld.in"), os.path.join(d, "rays.in"))
        env = dict(os.environ, RAYS_DUMP_FILE="none", RAYS_DUMP_RESULTS_LD="1")
        subprocess.run([REF], cwd=d, env=env, check=True, stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(d, "run_results.ld4"), os.path.join(out_root, LD_FIXTURE))
    print("run_results.gold_slab4_ld")


def deposition_file_fixture(out_root):
    """tests/golden/deposition_profiles.gaxi: the reference post-processor's own list-directed profile file
    (write_deposition_profiles_LD, deposition_profiles_m.f90:296-331) for configs/gold_axisym64_eqdsk_damp_rk4.in --
    data for rays_amd/results.py: write_deposition_profiles_LD (SURVEY 8(f) f2 -> files)."""
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(os.path.join(ROOT, "configs", "gold_axisym64_eqdsk_damp_rk4.in"), os.path.join(d, "rays.in"))
        for f in os.listdir(os.path.join(ROOT, "configs")):
            if f.endswith(".geqdsk"):
                shutil.copy(os.path.join(ROOT, "configs", f), d)
        env = dict(os.environ, RAYS_DUMP_FILE="dump.bin", RAYS_DUMP_DEPOSITION="dep.bin", RAYS_DUMP_DEPOSITION_LD="1")
        subprocess.run([REF], cwd=d, env=env, check=True, stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(d, "deposition_profiles.gaxi"), os.path.join(out_root, DEP_LD_FIXTURE))
    print("deposition_profiles.gaxi")


def same_bits(a, b):
    """Two arrays of a fixture: same dtype, shape and bytes (floats compared through their bit patterns)."""
    a, b = np.asarray(a), np.asarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def merge_tables(merged, new, where):
    """Union of the host tables of the eqdsk cases; an entry two cases both dump must be the same bits."""
    for k, v in new.items():
        v = np.asarray(v)
        if k in merged:
            if not same_bits(merged[k], v):
                sys.exit(f"{where}: host table '{k}' differs from the one another eqdsk case dumped")
        else:
            merged[k] = v


def generate(out_root, only=()):
    """Cut the fixtures `only` (all when empty) into <out_root>/tests/golden and the merged host tables into
    <out_root>/configs."""
    only = set(only)
    os.makedirs(os.path.join(out_root, "tests", "golden"), exist_ok=True)
    os.makedirs(os.path.join(out_root, "configs"), exist_ok=True)
    if not only or "run_results.gold_slab4_ld" in only:
        results_file_fixture(out_root)
    if not only or "deposition_profiles.gaxi" in only:
        deposition_file_fixture(out_root)
    host_tabs = {}
    for name, cfg, subset, stride, nprobe in CASES:
        if only and name not in only:
            continue
        check_deterministic(cfg)
        with tempfile.TemporaryDirectory() as d:
            shutil.copy(os.path.join(ROOT, "configs", cfg), os.path.join(d, "rays.in"))
            for f in os.listdir(os.path.join(ROOT, "configs")):
                if f.endswith(".geqdsk") or f.startswith("ray_init_"):   # (file_input_ray_init reads ray_init_<run_label>.in)
                    shutil.copy(os.path.join(ROOT, "configs", f), d)
            env = dict(os.environ, RAYS_DUMP_FILE="dump.bin", RAYS_DUMP_PROBE=str(stride),
                       RAYS_DUMP_AXISYM="axisym.bin", RAYS_DUMP_DEPOSITION="dep.bin")
            subprocess.run([REF], cwd=d, env=env, check=True, stdout=subprocess.DEVNULL)
            ref = read_dump(os.path.join(d, "dump.bin"))
            axi = read_axisym_tables(os.path.join(d, "axisym.bin")) \
                if os.path.exists(os.path.join(d, "axisym.bin")) else None
            dep = read_deposition(os.path.join(d, "dep.bin")) \
                if os.path.exists(os.path.join(d, "dep.bin")) else None
        idx = np.arange(ref["nray"]) if subset is None else np.array(subset)
        npts = ref["npoints"][idx]
        keep = int(npts.max())
        out = dict(
            config=np.array(cfg), ray_index=idx.astype(np.int32), nray_full=np.int32(ref["nray"]),
            npoints_full=ref["npoints"].astype(np.int32),
            stop_flag_full=np.array(ref["stop_flag"]),
            rvec0=ref["rvec0"][idx], rindex_vec0=ref["rindex_vec0"][idx],
            rvec0_full=ref["rvec0"], rindex_vec0_full=ref["rindex_vec0"],
            npoints=npts.astype(np.int32), stop_flag=np.array([ref["stop_flag"][i] for i in idx]),
            ray_vec=ref["ray_vec"][idx, :keep, :].copy(), residual=ref["residual"][idx, :keep].copy(),
            end_ray_vec=ref["end_ray_vec"][idx],
            consts=np.array([ref[k] for k in ("omgrf", "k0", "clight", "eps0")]),
            qs=ref["qs"], ms=ref["ms"], n0s=ref["n0s"], t0s=ref["t0s"],
            psiB=np.float64(ref["solovev"]["psiB"]),
        )
        # beyond npoints the reference arrays are zero (ray_results_m.f90:154-164)
        for r, n in enumerate(npts):
            assert not ref["ray_vec"][idx[r], n:, :].any() and not ref["residual"][idx[r], n:].any()
        if axi is not None:
            for k, v in axi.items():
                out["axi_" + k] = np.asarray(v)
            # the same tables next to the eqdsk file, for the Python host (RaysRun.from_namelist): merged over
            # all eqdsk cases below (each namelist builds only the profile splines it selects)
            if len(axi["r_grid"]) and "psi_fspl" in axi:   # (analytic magnetics: profile tables only; bilinear: Python reads the eqdsk)
                # a spline this namelist did not select is dumped empty: not an entry of the union
                tabs = {k: np.asarray(v) for k, v in axi.items() if np.ndim(v) == 0 or np.size(v)}
                if dep is not None and "rho_grid" in dep:   # rho(psiN) spline, for the Ptotal_rho profile
                    tabs.update(rho_grid=dep["rho_grid"], rho_fspl=dep["rho_fspl"])
                merge_tables(host_tabs.setdefault(eqdsk_of(cfg), {}), tabs, name)
        if dep is not None:
            # deposition profiles of the FULL fan + the full-fan trajectories they
            # are binned from (v(1:3), v(8)) so the device binner can be checked without a re-trace
            out["dep_n_bins"] = np.int32(dep["n_bins"])
            out["dep_power"] = dep["power"]
            out["dep_names"] = np.array(dep["names"])
            out["dep_work"] = np.stack(dep["work"])          # [profile][nray][n_bins]
            out["dep_profile"] = np.stack(dep["profile"])
            out["dep_q_sum"] = np.array(dep["q_sum"])
            if "rho_grid" in dep:
                out["dep_rho_grid"], out["dep_rho_fspl"] = dep["rho_grid"], dep["rho_fspl"]
            keep_full = int(ref["npoints"].max())
            out["dep_ray_vec_full"] = ref["ray_vec"][:, :keep_full, :].copy()
        if stride and "probes" in ref:
            pr = ref["probes"]
            sel = np.linspace(0, len(pr) - 1, min(nprobe, len(pr))).astype(int)
            out["probes"] = pr[sel]
        path = os.path.join(out_root, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: nray={ref['nray']} kept={len(idx)} maxpts={keep} "
              f"steps={int((ref['npoints'] - 1).sum())} -> {os.path.getsize(path) / 1e3:.0f} kB")
    for eqdsk, tabs in host_tabs.items():
        dst = os.path.join(out_root, tables_path(eqdsk))
        if only and os.path.exists(os.path.join(ROOT, tables_path(eqdsk))):
            # a partial regeneration keeps what the cases not run contributed (same rule: overlaps must agree)
            z = np.load(os.path.join(ROOT, tables_path(eqdsk)))
            merge_tables(tabs, {k: z[k] for k in z.files}, "committed " + tables_path(eqdsk))
        np.savez_compressed(dst, **tabs)
        print(f"{tables_path(eqdsk)}: {sorted(tabs)}")


def ld_records(path):
    """Lines of a list-directed results file without the records that differ from run to run: the date and the
    wall times (`date_vector`, `total_trace_time`, `ray_trace_time`; ray_results_m.f90:371-396)."""
    volatile = {"date_vector", "total_trace_time", "ray_trace_time"}
    names = {"RAYS_run_label", "date_vector", "number_of_rays", "max_number_of_points", "dim_v_vector", "npoints",
             "total_trace_time", "initial_ray_power", "ray_trace_time", "end_ray_parameter", "end_residuals",
             "max_residuals", "start_ray_vec", "end_ray_vec", "residual", "ray_vec", "ray_stop_flag"}
    keep, skipping = [], False
    for line in open(path):
        t = line.strip()
        if t in names:
            skipping = t in volatile
        if not skipping:
            keep.append(line.rstrip())
    return keep


def check():
    """Regenerate everything into a scratch directory and compare with the committed files bit for bit.
    Returns the list of differences (empty = the committed tree is what this script produces)."""
    diffs = []
    with tempfile.TemporaryDirectory() as d:
        generate(d)
        for name in [c[0] + ".npz" for c in CASES]:
            a, b = np.load(os.path.join(d, "tests", "golden", name)), np.load(os.path.join(ROOT, "tests", "golden", name))
            if set(a.files) != set(b.files):
                diffs.append(f"{name}: keys differ: only regenerated {sorted(set(a.files) - set(b.files))}, "
                             f"only committed {sorted(set(b.files) - set(a.files))}")
            diffs += [f"{name}: '{k}' differs" for k in sorted(set(a.files) & set(b.files)) if not same_bits(a[k], b[k])]
        for T in (tables_path(e) for e in EQDSK_FILES):
            a, b = np.load(os.path.join(d, T)), np.load(os.path.join(ROOT, T))
            if set(a.files) != set(b.files):
                diffs.append(f"{T}: keys differ: regenerated {sorted(a.files)}, committed {sorted(b.files)}")
            diffs += [f"{T}: '{k}' differs" for k in sorted(set(a.files) & set(b.files)) if not same_bits(a[k], b[k])]
        if ld_records(os.path.join(d, LD_FIXTURE)) != ld_records(os.path.join(ROOT, LD_FIXTURE)):
            diffs.append(f"{LD_FIXTURE}: differs beyond its date / wall-time records")
        if open(os.path.join(d, DEP_LD_FIXTURE)).read() != open(os.path.join(ROOT, DEP_LD_FIXTURE)).read():
            diffs.append(f"{DEP_LD_FIXTURE}: differs")
    return diffs


def main():
    args = sys.argv[1:]
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/rays_ref_dump missing: run `bash oracle/build_ref.sh` first")
    if "--check" in args:
        diffs = check()
        for x in diffs:
            print("DIFF", x)
        print("make_golden --check:", "committed fixtures are what this script produces" if not diffs
              else f"{len(diffs)} differences")
        sys.exit(1 if diffs else 0)
    generate(ROOT, args)  # optional: fixture names to (re)generate


if __name__ == "__main__":
    main()
