"""CPU tier: the on-disk result files (rays_amd/results.py) against the reference's own
write_results_LD output for configs/gold_slab4_results_ld.in (tests/golden/run_results.gold_slab4_ld,
written by the reference binary: tests/golden/make_golden.py)."""
import os
import struct

import numpy as np
import pytest

from rays_amd import results as R
from rays_amd.namelist import read_namelist
from rays_amd.params import params_from_namelist
from rays_amd.ray_init import initialize_ray_init
from tests.common import ROOT

REF_LD = os.path.join(ROOT, "tests", "golden", "run_results.gold_slab4_ld")
CFG = os.path.join(ROOT, "configs", "gold_slab4_results_ld.in")
ARRAYS = ("npoints", "initial_ray_power", "end_ray_parameter", "end_residuals", "max_residuals",
          "start_ray_vec", "end_ray_vec", "residual", "ray_vec")


class _Image:
    """the RayResults fields a RunResults is built from, taken from a parsed file"""

    def __init__(self, d):
        for k in ("ray_vec", "residual", "npoints", "end_ray_parameter", "end_residuals", "max_residuals",
                  "ray_stop_flag", "start_ray_vec", "end_ray_vec"):
            setattr(self, k, d[k])


def _from_file(d):
    return R.RunResults(_Image(d), d["initial_ray_power"], d["RAYS_run_label"], d["date_vector"],
                        d["total_trace_time"], d["ray_trace_time"])


def test_reader_parses_reference_file():
    d = R.read_results_LD(REF_LD)
    assert (d["RAYS_run_label"], d["number_of_rays"], d["max_number_of_points"], d["dim_v_vector"]) == ("ld4", 4, 31, 7)
    np.testing.assert_array_equal(d["npoints"], [31] * 4)
    assert d["ray_stop_flag"] == [" nstep > nstep_max".ljust(60)] * 4     # leading blank: ray_tracing.f90:152
    assert d["ray_vec"].shape == (4, 31, 7) and d["residual"].shape == (4, 31)
    np.testing.assert_array_equal(d["ray_vec"][:, 0, :], d["start_ray_vec"])
    np.testing.assert_array_equal(d["ray_vec"][:, 30, :], d["end_ray_vec"])
    np.testing.assert_array_equal(d["initial_ray_power"], [0.0625] * 4)  # weight divided by nray twice (App. A-11)


def test_list_directed_reals_follow_the_reference_build():
    """Tokens as amdflang writes them (values seen in reference files)."""
    for x, t in [(-0.08, "-8.E-02"), (0.0, "0."), (-0.6, "-.6"), (0.0625, "6.25E-02"), (3556.8295290699893, "3556.8295290699893"),
                 (1e15, "1.E+15"), (1e16, "1.E+16"), (100.0, "100."), (31.0, "31."), (0.1, ".1"), (1e-5, "1.E-05"),
                 (5e-324, "5.E-324"), (1.7976931348623157e308, "1.7976931348623157E+308"), (999999999999999.9, "999999999999999.9"),
                 (1.298845018062841e-02, "1.298845018062841E-02"), (float("inf"), "Inf"), (-float("inf"), "-Inf")]:
        assert R.ld_real(x) == t
    rng = np.random.default_rng(5)
    v = rng.uniform(-1, 1, 4000) * 10.0 ** rng.integers(-300, 300, 4000)
    for x in v:
        assert float(R.ld_real(x)) == x


def test_writer_reproduces_reference_file_layout_and_values(tmp_path):
    d = R.read_results_LD(REF_LD)
    out = str(tmp_path / "run_results.ld4")
    R.write_results_LD(out, _from_file(d))
    mine, ref = open(out).read().split("\n"), open(REF_LD).read().split("\n")
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):          # same records, same tokens per record, same widths
        assert len(a) == len(b) and len(a.split()) == len(b.split())
        for ta, tb in zip(a.split(), b.split()):
            if ta != tb:                 # a last digit may differ (see rays_amd/results.py); the value may not
                assert float(ta) == float(tb)
    e = R.read_results_LD(out)
    for k in ARRAYS + ("date_vector", "ray_trace_time"):
        np.testing.assert_array_equal(e[k], d[k])
    assert e["ray_stop_flag"] == d["ray_stop_flag"] and e["total_trace_time"] == d["total_trace_time"]


def test_netcdf_file_has_the_reference_schema(tmp_path):
    d = R.read_results_LD(REF_LD)
    d["npoints"] = np.array([31, 12, 1, 20], dtype=np.int32)       # ragged: the file is cut to maxval(npoints)
    r = _from_file(d)
    r.max_number_of_points = 31
    out = str(tmp_path / "run_results.ld4.nc")
    R.write_results_NC(out, r)
    assert open(out, "rb").read(4) == b"CDF\x01"                    # classic format (nf90_clobber)
    from scipy.io import netcdf_file
    with netcdf_file(out, "r", mmap=False) as f:
        assert {k: v for k, v in f.dimensions.items()} == dict(number_of_rays=4, max_number_of_points=31, dim_v_vector=7, d8=8, d60=60)
        want = dict(date_vector=("i", ("d8",)), ray_vec=("d", ("number_of_rays", "max_number_of_points", "dim_v_vector")),
                    residual=("d", ("number_of_rays", "max_number_of_points")), npoints=("i", ("number_of_rays",)),
                    initial_ray_power=("f", ("number_of_rays",)), ray_trace_time=("f", ("number_of_rays",)),
                    end_residuals=("f", ("number_of_rays",)), max_residuals=("f", ("number_of_rays",)),
                    end_ray_parameter=("f", ("number_of_rays",)), start_ray_vec=("f", ("number_of_rays", "dim_v_vector")),
                    end_ray_vec=("f", ("number_of_rays", "dim_v_vector")), ray_stop_flag=("c", ("number_of_rays", "d60")),
                    total_trace_time=("f", ()))
        assert set(f.variables) == set(want)
        for k, (t, dims) in want.items():
            assert (f.variables[k].typecode(), f.variables[k].dimensions) == (t, dims), k
    d["npoints"][0] = 17
    r = _from_file(d)
    R.write_results_NC(out, r)
    n = R.read_results_NC(out)
    assert n["max_number_of_points"] == 20 and n["ray_vec"].shape == (4, 20, 7)
    np.testing.assert_array_equal(n["ray_vec"], d["ray_vec"][:, :20, :])
    np.testing.assert_array_equal(n["residual"], d["residual"][:, :20])
    np.testing.assert_array_equal(n["end_ray_vec"], d["end_ray_vec"].astype(np.float32))
    assert n["ray_stop_flag"] == d["ray_stop_flag"] and n["RAYS_run_label"] == "ld4"


def test_kernel_source_on_host_writes_the_reference_file(tmp_path):
    """namelist -> ray launcher -> the product kernel source (host emulation) -> write_results_LD equals
    the reference's file in every array (dates and wall times aside)."""
    from rays_amd.trace import RayResults
    from tests import emul_lib
    nml = read_namelist(CFG)
    p = params_from_namelist(nml, None)
    r0, n0, w = initialize_ray_init(p, nml, None)
    res = RayResults(**emul_lib.trace(p, r0, n0))
    out = str(tmp_path / "run_results.ld4")
    R.write_results_LD(out, R.RunResults(res, w, "ld4"))
    mine, ref = R.read_results_LD(out), R.read_results_LD(REF_LD)
    for k in ARRAYS:
        np.testing.assert_array_equal(mine[k], ref[k], err_msg=k)
    assert mine["ray_stop_flag"] == ref["ray_stop_flag"]


# ---- run_results.<label>.nc: the file's STRUCTURE against the reference writer ------------------------------------
def _parse_classic_netcdf_header(path):
    """NetCDF classic (CDF-1 / CDF-2) header, parsed from the bytes (no NetCDF library): dimensions in
    definition order, global attributes, variables in definition order with type and dimension names."""
    b = open(path, "rb").read()
    assert b[:3] == b"CDF" and b[3] in (1, 2), "not a NetCDF classic file"
    off64 = b[3] == 2
    pos = 4

    def u32():
        nonlocal pos
        v = struct.unpack_from(">I", b, pos)[0]
        pos += 4
        return v

    def name():
        nonlocal pos
        n = u32()
        s = b[pos:pos + n].decode()
        pos += (n + 3) // 4 * 4
        return s

    TYPES = {1: "byte", 2: "char", 3: "short", 4: "int", 5: "float", 6: "double"}
    SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 4, 6: 8}

    def att_list():
        nonlocal pos
        tag, n = u32(), u32()
        assert (tag, n) == (0, 0) or tag == 0x0C
        out = {}
        for _ in range(n):
            k = name()
            t, m = u32(), u32()
            raw = b[pos:pos + m * SIZES[t]]
            pos += (m * SIZES[t] + 3) // 4 * 4
            out[k] = (TYPES[t], raw)
        return out

    numrecs = u32()
    tag, n = u32(), u32()
    assert tag == 0x0A
    dims = []
    for _ in range(n):
        k = name()
        dims.append((k, u32()))
    gatts = att_list()
    tag, n = u32(), u32()
    assert tag == 0x0B
    variables = []
    for _ in range(n):
        k = name()
        nd = u32()
        ids = [u32() for _ in range(nd)]
        att_list()
        t = u32()
        u32()  # vsize
        pos += 8 if off64 else 4  # begin
        variables.append((k, TYPES[t], tuple(dims[i][0] for i in ids)))
    return dict(numrecs=numrecs, dims=dims, gatts=gatts, vars=variables)


def test_netcdf_file_structure_is_the_reference_writers(tmp_path):
    """write_results_NC (ray_results_m.f90:171-249): dimension names, sizes and definition order (:205-209), variable
    names, NetCDF types and dimension lists in definition order (:212-224; the Fortran dimension lists are
    reversed in the file's C order), the global attribute (:227), max_number_of_points = maxval(npoints) (:202)."""
    d = R.read_results_LD(REF_LD)
    rr = _from_file(d)
    path = str(tmp_path / "run_results.ld4.nc")
    R.write_results_NC(path, rr)
    h = _parse_classic_netcdf_header(path)
    nray, npt, nv = 4, int(d["npoints"].max()), 7
    assert h["numrecs"] == 0   # no unlimited dimension
    assert h["dims"] == [("number_of_rays", nray), ("max_number_of_points", npt), ("dim_v_vector", nv), ("d8", 8),
                         ("d60", 60)]
    Rr, P, V = "number_of_rays", "max_number_of_points", "dim_v_vector"
    assert h["vars"] == [
        ("date_vector", "int", ("d8",)),
        ("ray_vec", "double", (Rr, P, V)),              # Fortran [dim_v_vector, max_number_of_points, number_of_rays]
        ("residual", "double", (Rr, P)),
        ("npoints", "int", (Rr,)),
        ("initial_ray_power", "float", (Rr,)),
        ("ray_trace_time", "float", (Rr,)),
        ("end_residuals", "float", (Rr,)),
        ("max_residuals", "float", (Rr,)),
        ("end_ray_parameter", "float", (Rr,)),
        ("start_ray_vec", "float", (Rr, V)),
        ("end_ray_vec", "float", (Rr, V)),
        ("ray_stop_flag", "char", (Rr, "d60")),
        ("total_trace_time", "float", ()),
    ]
    assert list(h["gatts"]) == ["RAYS_run_label"] and h["gatts"]["RAYS_run_label"][0] == "char"
    assert h["gatts"]["RAYS_run_label"][1].decode().strip() == "ld4"
    # and the payload, read back independently of the writer's own reader conventions
    nc = R.read_results_NC(path)
    np.testing.assert_array_equal(nc["ray_vec"], d["ray_vec"][:, :npt])
    np.testing.assert_array_equal(nc["residual"], d["residual"][:, :npt])
    np.testing.assert_array_equal(nc["end_ray_vec"], d["end_ray_vec"].astype(np.float32))
    assert nc["ray_stop_flag"] == d["ray_stop_flag"]


def test_deposition_profile_file_reproduces_the_reference_post_processors(tmp_path):
    """rays_amd/results.py: write_deposition_profiles_LD against the reference post-processor's own
    `deposition_profiles.<run_label>` for configs/gold_axisym64_eqdsk_damp_rk4.in (write_deposition_profiles_LD,
    deposition_profiles_m.f90:296-331; tests/golden/deposition_profiles.gaxi, written by the reference binary):
    profiles, Q_sum and bin edges from the fixture / the grid formula -> the same records and the same binary64 values."""
    ref = os.path.join(ROOT, "tests", "golden", "deposition_profiles.gaxi")
    g = np.load(os.path.join(ROOT, "tests", "golden", "gold_axisym64_eqdsk_damp_rk4.npz"))
    n_bins = int(g["dep_n_bins"])
    names = [str(n).strip() for n in g["dep_names"]]
    assert names == ["Ptotal_psi", "Ptotal_rho"]
    profiles = [dict(profile_name=n, grid_name=n.split("_")[1], profile=g["dep_profile"][i], Q_sum=float(g["dep_q_sum"][i]),
                     grid=R.deposition_grid(0.0, 1.0, n_bins)) for i, n in enumerate(names)]   # :176-177: psiN, rho in 0..1
    out = tmp_path / "deposition_profiles.gaxi"
    R.write_deposition_profiles_LD(str(out), profiles)
    # same records, same number of values per record, every value the same binary64 (where a value needs all 17
    # digits the compiler's runtime and ld_real may pick different ones of the strings that read back to it:
    # .10119588148219836 / ...837 are the same double)
    mine, theirs = out.read_text().split("\n"), open(ref).read().split("\n")
    assert len(mine) == len(theirs)
    for a, b in zip(mine, theirs):
        if "=" in b or "Ptotal_total_deposition" in b:
            assert a == b                          # name records: byte for byte (names padded to character(len=20))
        else:
            ta, tb = a.split(), b.split()
            assert len(ta) == len(tb) and [float(x) for x in ta] == [float(x) for x in tb], (a, b)
    # Q_sum is the ordered sum of the profile (deposition_profiles_m.f90:251)
    for pr in profiles:
        s = 0.0
        for v in pr["profile"]:
            s = s + float(v)
        assert s == pr["Q_sum"]


def test_deposition_profile_netcdf_file_has_the_reference_writers_structure(tmp_path):
    """rays_amd/results.py: write_deposition_profiles_NC against the definitions of the reference's
    write_deposition_profiles_NC (deposition_profiles_m.f90:374-391: n_profiles UNLIMITED, n_bins, n_bins_p1, d20;
    Q_sum, n_bins, grid_min, grid_max, profile_name, grid_name, grid, profile in that order; RAYS_run_label and
    date_vector as global attributes).  No NetCDF library in this image writes the reference's file, so the header is
    parsed from the bytes (names, order, types, record dimension) and the values are read back with scipy."""
    from scipy.io import netcdf_file
    g = np.load(os.path.join(ROOT, "tests", "golden", "gold_axisym64_eqdsk_damp_rk4.npz"))
    n_bins = int(g["dep_n_bins"])
    names = [str(n).strip() for n in g["dep_names"]]
    profiles = [dict(profile_name=n, grid_name=n.split("_")[1], profile=g["dep_profile"][i], Q_sum=float(g["dep_q_sum"][i]),
                     grid=R.deposition_grid(0.0, 1.0, n_bins), grid_min=0.0, grid_max=1.0) for i, n in enumerate(names)]
    out = str(tmp_path / "deposition_profiles.gaxi.nc")
    R.write_deposition_profiles_NC(out, profiles, run_label="gaxi", date_vector=[2026, 10, 4, 0, 1, 2, 3, 4])
    b = open(out, "rb").read()
    assert b[:4] == b"CDF\x01" and struct.unpack(">I", b[4:8])[0] == 2            # two records = two profiles
    pos = 8

    def u32():
        nonlocal pos
        v = struct.unpack(">I", b[pos:pos + 4])[0]
        pos += 4
        return v

    def name():
        nonlocal pos
        n = u32()
        s = b[pos:pos + n].decode()
        pos += n + (-n % 4)
        return s

    assert u32() == 0x0A
    dims = [(name(), u32()) for _ in range(u32())]
    assert dims == [("n_profiles", 0), ("n_bins", n_bins), ("n_bins_p1", n_bins + 1), ("d20", 20)]   # 0 = the record dimension
    assert u32() == 0x0C
    atts = []
    for _ in range(u32()):
        an, typ, cnt = name(), u32(), u32()
        size = {2: 1, 4: 4}[typ] * cnt
        atts.append((an, typ, cnt))
        pos += size + (-size % 4)
    assert atts == [("RAYS_run_label", 2, 4), ("date_vector", 4, 8)]
    assert u32() == 0x0B
    got = []
    for _ in range(u32()):
        vn = name()
        vd = [u32() for _ in range(u32())]
        assert u32() == 0 and u32() == 0            # no variable attributes
        typ, vsize, begin = u32(), u32(), u32()
        got.append((vn, typ, vd))
    D, I, C_ = 6, 4, 2
    assert got == [("Q_sum", D, [0]), ("n_bins", I, [0]), ("grid_min", D, [0]), ("grid_max", D, [0]),
                   ("profile_name", C_, [0, 3]), ("grid_name", C_, [0, 3]), ("grid", D, [0, 2]), ("profile", D, [0, 1])]
    with netcdf_file(out, "r", mmap=False) as f:
        assert f.variables["profile"].isrec and f.dimensions["n_profiles"] is None
        np.testing.assert_array_equal(np.array(f.variables["profile"].data), g["dep_profile"])
        np.testing.assert_array_equal(np.array(f.variables["Q_sum"].data), g["dep_q_sum"])
        np.testing.assert_array_equal(np.array(f.variables["grid"].data)[1], R.deposition_grid(0.0, 1.0, n_bins))
        assert b"".join(np.array(f.variables["profile_name"].data)[1]).decode().strip() == "Ptotal_rho"
        np.testing.assert_array_equal(np.array(f.variables["n_bins"].data), [n_bins, n_bins])
        assert list(f.date_vector) == [2026, 10, 4, 0, 1, 2, 3, 4]
