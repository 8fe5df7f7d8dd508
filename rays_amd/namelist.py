"""Minimal Fortran-namelist reader for RAYS input files (`rays.in`).

The reference reads one namelist group per module from the same file
(e.g. RAYS_project/RAYS_lib/ode_m.f90:127-133); this reader returns
``{group: {key: value}}`` with lower-cased group/key names.  Indexed assignments
(``eta(1)=1.``) become ``{key: {index: value}}``; repeat counts (``2*'zero'``) expand
to lists.  Later assignments override earlier ones, like a Fortran namelist read.
"""
from __future__ import annotations

import re
from typing import Any, Dict

_TOKEN = re.compile(
    r"""\s*(?:
        (?P<str>'(?:[^']|'')*'|"(?:[^"]|"")*")      # quoted string
      | (?P<word>[^\s,='"/]+)                          # bare token
      | (?P<eq>=)
      | (?P<comma>,)
      | (?P<slash>/)
    )""",
    re.X,
)


def _strip_comments(text: str) -> str:
    out = []
    for line in text.splitlines():
        buf, q = [], None
        for ch in line:
            if q:
                buf.append(ch)
                if ch == q:
                    q = None
            elif ch in "'\"":
                q = ch
                buf.append(ch)
            elif ch == "!":
                break
            else:
                buf.append(ch)
        out.append("".join(buf))
    return "\n".join(out)


def _scalar(tok: str) -> Any:
    t = tok.strip()
    lo = t.lower()
    if lo in (".true.", "t", ".t.", "true"):
        return True
    if lo in (".false.", "f", ".f.", "false"):
        return False
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(lo.replace("d", "e"))
    except ValueError:
        return t


def _value(tok: str, is_str: bool) -> Any:
    if is_str:
        q = tok[0]
        return tok[1:-1].replace(q + q, q)
    return _scalar(tok)


def parse_namelist(text: str) -> Dict[str, Dict[str, Any]]:
    text = _strip_comments(text)
    groups: Dict[str, Dict[str, Any]] = {}
    pos = 0
    while True:
        m = re.compile(r"[&$]\s*([A-Za-z_]\w*)").search(text, pos)
        if not m:
            break
        gname = m.group(1).lower()
        pos = m.end()
        group = groups.setdefault(gname, {})
        toks = []
        while True:
            tm = _TOKEN.match(text, pos)
            if not tm:
                break
            pos = tm.end()
            if tm.group("slash"):
                break
            if tm.group("str") is not None:
                toks.append(("str", tm.group("str")))
            elif tm.group("word") is not None:
                w = tm.group("word")
                if w.lower() in ("&end", "$end"):
                    break
                toks.append(("word", w))
            elif tm.group("eq"):
                toks.append(("eq", "="))
        # split into assignments: a `word` followed by `eq` starts a new key
        i = 0
        key, idx, vals = None, None, []

        def flush():
            if key is None:
                return
            v = vals[0] if len(vals) == 1 else list(vals)
            if idx is not None:
                d = group.setdefault(key, {})
                if not isinstance(d, dict):
                    d = {}
                    group[key] = d
                for off, item in enumerate(vals):
                    d[idx + off] = item
            else:
                group[key] = v

        while i < len(toks):
            kind, tok = toks[i]
            if kind == "word" and i + 1 < len(toks) and toks[i + 1][0] == "eq":
                flush()
                km = re.match(r"([A-Za-z_]\w*)(?:\((-?\d+)\))?$", tok)
                if not km:
                    raise ValueError(f"bad namelist key {tok!r} in group {gname}")
                key = km.group(1).lower()
                idx = int(km.group(2)) if km.group(2) is not None else None
                vals = []
                i += 2
                continue
            if kind == "word" and "*" in tok and not tok.startswith("*"):
                # repeat count: n*value (value may be the following string token)
                n, _, rest = tok.partition("*")
                if rest == "" and i + 1 < len(toks) and toks[i + 1][0] == "str":
                    vals.extend([_value(toks[i + 1][1], True)] * int(n))
                    i += 2
                    continue
                vals.extend([_scalar(rest)] * int(n))
                i += 1
                continue
            vals.append(_value(tok, kind == "str"))
            i += 1
        flush()
    return groups


def read_namelist(path: str) -> Dict[str, Dict[str, Any]]:
    with open(path, "r") as f:
        return parse_namelist(f.read())
