"""On-disk images of the ray_results_m arrays: the two files the reference writes
in finalize_run, so that post_process_RAYS and graphics_RAYS consume a GPU run unchanged.

    write_results_LD  ->  run_results.<label>      list-directed text  (ray_results_m.f90:365-420)
    write_results_NC  ->  run_results.<label>.nc   NetCDF classic      (ray_results_m.f90:171-249)
    read_results_LD / read_results_NC              the matching readers (425-600 / 253-361)
    write_deposition_profiles_LD / _NC  ->  deposition_profiles.<label>[.nc]   the post-processor's profile files
                                            (post_process_lib/deposition_profiles_m.f90:296-331, 336-420)

The text writer reproduces the record layout the reference's own build produces (flang
list-directed output: one leading blank per record, records of at most 79 columns, a shortest
round-trip decimal for every real, F editing for decimal exponents 0..15 and 1P E editing
otherwise, character arrays run together and cut at the record length).  Every value reads back to
the same binary64 as from the reference's file; the text itself can differ in a last digit, because
where two decimals of the shortest length both round-trip flang does not always print the nearer
one (about 3 % of random values) and this writer does.  The NetCDF file is the classic format `nf90_create(..., nf90_clobber)`
gives, written through scipy.io (no NetCDF library in this image); ray_vec and residual are cut to
maxval(npoints) like the reference does (ray_results_m.f90:204).

With the Fortran host (fortran/trace_rays_hip.f90) none of this is needed: the reference's own
writers run on the arrays the shim filled.  This module serves the Python mirror.
"""
from __future__ import annotations

import datetime
import decimal
import math
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np

_RECL = 79          # flang's default record length for list-directed output
_FLAG_LEN = 60      # character(len=60) :: ray_stop_flag(:)   (ray_results_m.f90:56)
_LD_ORDER = ("npoints", "total_trace_time", "initial_ray_power", "ray_trace_time", "end_ray_parameter",
             "end_residuals", "max_residuals", "ray_stop_flag", "start_ray_vec", "end_ray_vec", "residual",
             "ray_vec")


def ld_real(x: float) -> str:
    """One real in list-directed form: shortest digits that round-trip binary64; value =
    0.d1d2... x 10**expo is written with F editing when 0 <= expo <= 15, else as d.ddE+xx."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    sign = "-" if math.copysign(1.0, x) < 0 else ""
    if x == 0.0:
        return sign + "0."
    t = decimal.Decimal(repr(abs(x))).as_tuple()
    digits = "".join(map(str, t.digits)).lstrip("0")
    exp10 = t.exponent
    stripped = digits.rstrip("0")
    exp10 += len(digits) - len(stripped)
    digits = stripped
    expo = len(digits) + exp10
    if 0 <= expo <= 15:
        if len(digits) <= expo:
            return sign + digits + "0" * (expo - len(digits)) + "."
        return sign + digits[:expo] + "." + digits[expo:]
    e = expo - 1
    return f"{sign}{digits[0]}.{digits[1:]}E{'-' if e < 0 else '+'}{abs(e):02d}"


def _records(tokens: Iterable[str]) -> List[str]:
    out, line = [], ""
    for t in tokens:
        if line and len(line) + 1 + len(t) > _RECL:
            out.append(line)
            line = ""
        line += " " + t
    out.append(line if line else " ")
    return out


def _char_records(strings: Sequence[str], width: int) -> List[str]:
    blob = "".join(s[:width].ljust(width) for s in strings)
    n = _RECL - 1
    return [" " + blob[i:i + n] for i in range(0, max(len(blob), 1), n)]


class RunResults:
    """Image of the reference's `run_results` derived type (ray_results_m.f90:60-91): the module
    arrays of one run plus its label and date."""

    def __init__(self, results, initial_ray_power=None, run_label: str = "", date_vector=None,
                 total_trace_time: Optional[float] = None, ray_trace_time=None):
        nray = len(results.npoints)
        self.RAYS_run_label = str(run_label)
        if date_vector is None:
            now = datetime.datetime.now().astimezone()
            off = now.utcoffset()
            date_vector = [now.year, now.month, now.day, int(off.total_seconds() // 60) if off else 0,
                           now.hour, now.minute, now.second, now.microsecond // 1000]
        self.date_vector = np.asarray(date_vector, dtype=np.int32)
        self.ray_vec = np.ascontiguousarray(results.ray_vec, dtype=np.float64)
        self.residual = np.ascontiguousarray(results.residual, dtype=np.float64)
        self.npoints = np.ascontiguousarray(results.npoints, dtype=np.int32)
        self.number_of_rays = nray
        self.max_number_of_points = self.ray_vec.shape[1]
        self.dim_v_vector = self.ray_vec.shape[2]
        z = np.zeros(nray)
        self.initial_ray_power = z.copy() if initial_ray_power is None else np.asarray(initial_ray_power, dtype=np.float64)
        # the device traces all rays at once: there is no per-ray wall time (left at the zero fill of
        # ray_results_m.f90:158); total_trace_time is the whole call
        self.ray_trace_time = z.copy() if ray_trace_time is None else np.asarray(ray_trace_time, dtype=np.float64)
        self.total_trace_time = float(getattr(results, "elapsed_s", 0.0) if total_trace_time is None else total_trace_time)
        self.end_ray_parameter = np.ascontiguousarray(results.end_ray_parameter, dtype=np.float64)
        self.end_residuals = np.ascontiguousarray(results.end_residuals, dtype=np.float64)
        self.max_residuals = np.ascontiguousarray(results.max_residuals, dtype=np.float64)
        self.ray_stop_flag = list(results.ray_stop_flag)
        self.start_ray_vec = np.ascontiguousarray(results.start_ray_vec, dtype=np.float64)
        self.end_ray_vec = np.ascontiguousarray(results.end_ray_vec, dtype=np.float64)


def write_results_LD(path: str, r: RunResults) -> None:
    """ray_results_m.f90:365-420: a name record then a value record (group) per variable, arrays in
    Fortran element order (= C order of the [nray][point][nv] images)."""
    reals = lambda a: _records(ld_real(v) for v in np.asarray(a, dtype=np.float64).ravel())
    ints = lambda a: _records(str(int(v)) for v in np.asarray(a).ravel())
    lines = [" RAYS_run_label"] + _char_records([r.RAYS_run_label], _FLAG_LEN)
    lines += [" date_vector"] + ints(r.date_vector)
    lines += [" number_of_rays"] + ints([r.number_of_rays])
    lines += [" max_number_of_points"] + ints([r.max_number_of_points])
    lines += [" dim_v_vector"] + ints([r.dim_v_vector])
    for name in _LD_ORDER:
        lines.append(" " + name)
        v = getattr(r, name)
        if name == "npoints":
            lines += ints(v)
        elif name == "ray_stop_flag":
            lines += _char_records(v, _FLAG_LEN)
        elif name == "total_trace_time":
            lines += reals([v])
        else:
            lines += reals(v)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def deposition_grid(grid_min: float, grid_max: float, n_bins: int) -> np.ndarray:
    """grid(1:n_bins+1) of a deposition profile (deposition_profiles_m.f90:156-159, 198-201):
    grid_min + delta*(i-1) with delta = (grid_max - grid_min)/real(n_bins)."""
    delta = (float(grid_max) - float(grid_min)) / float(np.float32(n_bins))
    return np.array([float(grid_min) + delta * i for i in range(n_bins + 1)])


def write_deposition_profiles_LD(path: str, profiles: Sequence[Dict[str, Any]]) -> None:
    """write_deposition_profiles_LD (post_process_lib/deposition_profiles_m.f90:296-331): the file
    `deposition_profiles.<run_label>` that post_process_RAYS / graphics_RAYS read.  Per profile: a name record,
    the profile, a grid-name record, the grid (n_bins + 1 edges), 'Ptotal_total_deposition' and Q_sum.
    profiles: dicts with profile_name, grid_name (character(len=20) in the reference), profile[n_bins],
    grid[n_bins+1] (deposition_grid), Q_sum -- e.g. the output of rays_hip_deposition."""
    reals = lambda a: _records(ld_real(v) for v in np.asarray(a, dtype=np.float64).ravel())
    lines: List[str] = []
    for pr in profiles:
        lines.append(" profile_name = " + str(pr["profile_name"])[:20].ljust(20))
        lines += reals(pr["profile"])
        lines.append(" grid_name = " + str(pr["grid_name"])[:20].ljust(20))
        lines += reals(pr["grid"])
        lines.append(" Ptotal_total_deposition")
        lines += reals([pr["Q_sum"]])
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def _ld_values(tok: str) -> List[str]:
    """One list-directed token, with the r*c repeat form some compilers use."""
    if "*" in tok:
        n, v = tok.split("*", 1)
        return [v] * int(n)
    return [tok]


def read_results_LD(path: str) -> Dict[str, Any]:
    """read_results_LD (ray_results_m.f90:425-600): name record, then as many values as the shapes read
    from the head of the file call for.  Returns the arrays in C order ([nray][point][nv])."""
    with open(path) as f:
        lines = f.read().split("\n")
    pos = 0

    def name(expect):
        nonlocal pos
        got = lines[pos].strip()
        pos += 1
        if got != expect:
            raise ValueError(f"read_results_LD: inconsistent variable name = {got!r} (expected {expect!r})")

    def numbers(count, conv):
        nonlocal pos
        vals: List[str] = []
        while len(vals) < count:
            for tok in lines[pos].replace(",", " ").split():
                vals += _ld_values(tok)
            pos += 1
        if len(vals) != count:
            raise ValueError(f"read_results_LD: {len(vals)} values where {count} were expected")
        return [conv(v.replace("D", "E").replace("d", "E")) for v in vals]

    def chars(count, width):
        nonlocal pos
        blob, need, n = "", count * width, _RECL - 1
        while len(blob) < need:          # records may have lost their trailing blanks
            blob += lines[pos][1:n + 1].ljust(min(n, need - len(blob)))
            pos += 1
        return [blob[i * width:(i + 1) * width] for i in range(count)]

    out: Dict[str, Any] = {}
    name("RAYS_run_label")
    out["RAYS_run_label"] = chars(1, _FLAG_LEN)[0].strip()
    name("date_vector")
    out["date_vector"] = np.array(numbers(8, int), dtype=np.int32)
    for k in ("number_of_rays", "max_number_of_points", "dim_v_vector"):
        name(k)
        out[k] = numbers(1, int)[0]
    nray, npt, nv = out["number_of_rays"], out["max_number_of_points"], out["dim_v_vector"]
    shapes = dict(total_trace_time=(), initial_ray_power=(nray,), ray_trace_time=(nray,), end_ray_parameter=(nray,),
                  end_residuals=(nray,), max_residuals=(nray,), start_ray_vec=(nray, nv), end_ray_vec=(nray, nv),
                  residual=(nray, npt), ray_vec=(nray, npt, nv))
    for k in _LD_ORDER:
        name(k)
        if k == "npoints":
            out[k] = np.array(numbers(nray, int), dtype=np.int32)
        elif k == "ray_stop_flag":
            out[k] = chars(nray, _FLAG_LEN)
        else:
            shp = shapes[k]
            a = np.array(numbers(int(np.prod(shp)) if shp else 1, float), dtype=np.float64)
            out[k] = float(a[0]) if shp == () else a.reshape(shp)
    return out


_NC_TYPE = {"char": (2, "S1", 1), "int": (4, ">i4", 4), "float": (5, ">f4", 4), "double": (6, ">f8", 8)}


def _write_netcdf_classic(path: str, dims, gatts, variables) -> None:
    """A NetCDF classic file (CDF-1; CDF-2 = 64-bit offsets when a variable starts beyond 2 GiB) with
    dimensions, global char attributes and fixed-size variables laid out IN DEFINITION ORDER, as the netCDF
    library lays out what the reference's write_results_NC defines (scipy's writer re-orders the variables).
    dims: [(name, size)]; gatts: [(name, text)]; variables: [(name, type, (dim names), array)]."""
    import struct

    def pad4(b: bytes) -> bytes:
        return b + b"\0" * (-len(b) % 4)

    def nm(sname: str) -> bytes:
        e = sname.encode()
        return struct.pack(">I", len(e)) + pad4(e)

    dim_index = {d[0]: i for i, d in enumerate(dims)}
    blobs = []
    for vname, typ, vdims, data in variables:
        code, dt, size = _NC_TYPE[typ]
        arr = np.asarray(data).astype(dt) if typ != "char" else np.asarray(data, dtype="S1")
        want = tuple(dict(dims)[d] for d in vdims)
        arr = np.ascontiguousarray(arr).reshape(want) if arr.size == int(np.prod(want, dtype=np.int64)) else arr
        if arr.shape != want:
            raise ValueError(f"{vname}: shape {arr.shape} != {want}")
        blobs.append(pad4(arr.tobytes()))
    for off_bytes in (4, 8):
        head = [b"CDF" + bytes([1 if off_bytes == 4 else 2]), struct.pack(">I", 0)]
        head.append(struct.pack(">II", 0x0A, len(dims)) if dims else struct.pack(">II", 0, 0))
        for dname, size in dims:
            head.append(nm(dname) + struct.pack(">I", size))
        head.append(struct.pack(">II", 0x0C, len(gatts)) if gatts else struct.pack(">II", 0, 0))
        for aname, text in gatts:
            e = text.encode()
            head.append(nm(aname) + struct.pack(">II", 2, len(e)) + pad4(e))
        head.append(struct.pack(">II", 0x0B, len(variables)) if variables else struct.pack(">II", 0, 0))
        var_heads = []
        for (vname, typ, vdims, _), blob in zip(variables, blobs):
            h = nm(vname) + struct.pack(">I", len(vdims)) + b"".join(struct.pack(">I", dim_index[d]) for d in vdims)
            h += struct.pack(">II", 0, 0)  # no variable attributes
            h += struct.pack(">I", _NC_TYPE[typ][0]) + struct.pack(">I", min(len(blob), 0xFFFFFFFF))
            var_heads.append(h)
        hlen = sum(len(x) for x in head) + sum(len(h) + off_bytes for h in var_heads)
        begins, pos = [], hlen
        for blob in blobs:
            begins.append(pos)
            pos += len(blob)
        if off_bytes == 8 or all(bg < (1 << 31) for bg in begins):
            break
    with open(path, "wb") as f:
        f.write(b"".join(head))
        for h, bg in zip(var_heads, begins):
            f.write(h + struct.pack(">I" if off_bytes == 4 else ">Q", bg))
        for blob in blobs:
            f.write(blob)


def _write_netcdf_classic_records(path: str, dims, numrecs: int, gatts, variables) -> None:
    """A NetCDF classic (CDF-1) file whose variables are all RECORD variables over the unlimited dimension dims[0]
    (size 0 in the header, `numrecs` records): the layout the netCDF library gives what the reference's
    write_deposition_profiles_NC defines.  gatts: [(name, text | int array)]; variables: [(name, type, (dim names,
    the record dimension first), array[numrecs, ...])].  Records are interleaved: record r of a variable starts at
    its `begin` + r * (sum of all variables' per-record sizes, each padded to four bytes)."""
    import struct

    def pad4(b: bytes) -> bytes:
        return b + b"\0" * (-len(b) % 4)

    def nm(sname: str) -> bytes:
        e = sname.encode()
        return struct.pack(">I", len(e)) + pad4(e)

    dim_index = {d[0]: i for i, d in enumerate(dims)}
    sizes = dict(dims)
    recs = []  # per variable: list of per-record blobs
    for vname, typ, vdims, data in variables:
        if not vdims or vdims[0] != dims[0][0]:
            raise ValueError(f"{vname}: not a record variable")
        code, dt, size = _NC_TYPE[typ]
        arr = np.asarray(data).astype(dt) if typ != "char" else np.asarray(data, dtype="S1")
        want = (numrecs,) + tuple(sizes[d] for d in vdims[1:])
        arr = np.ascontiguousarray(arr).reshape(want)
        recs.append([pad4(arr[r:r + 1].tobytes()) for r in range(numrecs)])   # (a slice keeps the big-endian dtype)
    vsizes = [len(x[0]) if x else 0 for x in recs]
    head = [b"CDF\x01", struct.pack(">I", numrecs), struct.pack(">II", 0x0A, len(dims))]
    for i, (dname, size) in enumerate(dims):
        head.append(nm(dname) + struct.pack(">I", 0 if i == 0 else size))
    head.append(struct.pack(">II", 0x0C, len(gatts)) if gatts else struct.pack(">II", 0, 0))
    for aname, val in gatts:
        if isinstance(val, str):
            e = val.encode()
            head.append(nm(aname) + struct.pack(">II", 2, len(e)) + pad4(e))
        else:
            a = np.asarray(val).astype(">i4")
            head.append(nm(aname) + struct.pack(">II", 4, a.size) + a.tobytes())
    head.append(struct.pack(">II", 0x0B, len(variables)))
    var_heads = []
    for (vname, typ, vdims, _), vs in zip(variables, vsizes):
        h = nm(vname) + struct.pack(">I", len(vdims)) + b"".join(struct.pack(">I", dim_index[d]) for d in vdims)
        h += struct.pack(">II", 0, 0) + struct.pack(">I", _NC_TYPE[typ][0]) + struct.pack(">I", vs)
        var_heads.append(h)
    hlen = sum(len(x) for x in head) + sum(len(h) + 4 for h in var_heads)
    begins, pos = [], hlen
    for vs in vsizes:
        begins.append(pos)
        pos += vs
    with open(path, "wb") as f:
        f.write(b"".join(head))
        for h, bg in zip(var_heads, begins):
            f.write(h + struct.pack(">I", bg))
        for r in range(numrecs):
            for x in recs:
                f.write(x[r])


def write_deposition_profiles_NC(path: str, profiles: Sequence[Dict[str, Any]], run_label: str = "", date_vector=None) -> None:
    """write_deposition_profiles_NC (post_process_lib/deposition_profiles_m.f90:336-420): `deposition_profiles.<label>.nc`
    with the reference's dimensions (n_profiles UNLIMITED, n_bins, n_bins_p1, d20; :374-377), its eight variables in its
    definition order and types (:380-387; Fortran dimension lists reversed into the file's C order) and its two global
    attributes (RAYS_run_label, date_vector; :390-391).  profiles: as for write_deposition_profiles_LD, plus grid_min,
    grid_max."""
    n = len(profiles)
    n_bins = len(np.asarray(profiles[0]["profile"])) if n else 0
    if date_vector is None:
        now = datetime.datetime.now().astimezone()
        off = now.utcoffset()
        date_vector = [now.year, now.month, now.day, int(off.total_seconds() // 60) if off else 0, now.hour, now.minute, now.second,
                       now.microsecond // 1000]
    P = "n_profiles"
    dims = [(P, 0), ("n_bins", n_bins), ("n_bins_p1", n_bins + 1), ("d20", 20)]
    name20 = lambda k: np.array([list(str(pr[k])[:20].ljust(20)) for pr in profiles], dtype="S1").reshape(n, 20)
    variables = [("Q_sum", "double", (P,), [pr["Q_sum"] for pr in profiles]),
                 ("n_bins", "int", (P,), [len(np.asarray(pr["profile"])) for pr in profiles]),
                 ("grid_min", "double", (P,), [pr["grid_min"] for pr in profiles]),
                 ("grid_max", "double", (P,), [pr["grid_max"] for pr in profiles]),
                 ("profile_name", "char", (P, "d20"), name20("profile_name")),
                 ("grid_name", "char", (P, "d20"), name20("grid_name")),
                 ("grid", "double", (P, "n_bins_p1"), [pr["grid"] for pr in profiles]),
                 ("profile", "double", (P, "n_bins"), [pr["profile"] for pr in profiles])]
    _write_netcdf_classic_records(path, dims, n, [("RAYS_run_label", str(run_label)), ("date_vector", np.asarray(date_vector, dtype=np.int32))],
                                  variables)


def write_results_NC(path: str, r: RunResults) -> None:
    """write_results_NC (ray_results_m.f90:171-249): the same dimensions (:205-209) and variables (:212-224) in the
    same definition order with the same NetCDF types (ray_vec, residual double; the per-ray summaries NF90_FLOAT;
    ray_stop_flag char [number_of_rays][60]; the Fortran dimension lists reversed into the file's C order), the
    global attribute RAYS_run_label (:227); ray_vec / residual cut to maxval(npoints) (:202)."""
    npt = int(r.npoints.max()) if r.number_of_rays else 0
    R, P, V = "number_of_rays", "max_number_of_points", "dim_v_vector"
    dims = [(R, r.number_of_rays), (P, npt), (V, r.dim_v_vector), ("d8", 8), ("d60", _FLAG_LEN)]
    flags = np.array([list(s[:_FLAG_LEN].ljust(_FLAG_LEN)) for s in r.ray_stop_flag], dtype="S1")
    variables = [("date_vector", "int", ("d8",), r.date_vector),
                 ("ray_vec", "double", (R, P, V), r.ray_vec[:, :npt, :]),
                 ("residual", "double", (R, P), r.residual[:, :npt]),
                 ("npoints", "int", (R,), r.npoints)]
    for k in ("initial_ray_power", "ray_trace_time", "end_residuals", "max_residuals", "end_ray_parameter"):
        variables.append((k, "float", (R,), getattr(r, k)))
    variables += [("start_ray_vec", "float", (R, V), r.start_ray_vec), ("end_ray_vec", "float", (R, V), r.end_ray_vec),
                  ("ray_stop_flag", "char", (R, "d60"), flags.reshape(r.number_of_rays, _FLAG_LEN)),
                  ("total_trace_time", "float", (), np.float32(r.total_trace_time))]
    _write_netcdf_classic(path, dims, [("RAYS_run_label", r.RAYS_run_label.ljust(_FLAG_LEN))], variables)


def read_results_NC(path: str) -> Dict[str, Any]:
    """read_results_instance_NC (ray_results_m.f90:253-361)."""
    from scipy.io import netcdf_file

    out: Dict[str, Any] = {}
    with netcdf_file(path, "r", mmap=False) as f:
        lab = f.RAYS_run_label
        out["RAYS_run_label"] = (lab.decode() if isinstance(lab, bytes) else str(lab)).strip()
        for d in ("number_of_rays", "max_number_of_points", "dim_v_vector"):
            out[d] = int(f.dimensions[d])
        for k, v in f.variables.items():
            a = np.array(v.data)
            if k == "ray_stop_flag":
                out[k] = [b"".join(row).decode() for row in a.reshape(out["number_of_rays"], _FLAG_LEN)]
            else:
                out[k] = a if a.ndim else a[()]
    return out
