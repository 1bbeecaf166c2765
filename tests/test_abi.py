"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/elastic_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

from elasticdiffusion_official_amd import _hip


def header_functions():
    text = open(os.path.join(_hip.INCLUDE, "elastic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\b(?:int64_t|int|const char\*)\s+(ed_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)


def test_library_builds_and_exports_every_declared_symbol():
    _hip.build_library()
    L = _hip.lib()
    declared = header_functions()
    names = [n for n, _ in declared]
    assert len(names) >= 13 and len(set(names)) == len(names)
    assert set(names) == set(_hip.SIGNATURES), "header and ctypes signature table disagree"
    for name, args in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
        n_args = 0 if args.strip() in ("", "void") else len([a for a in args.split(",") if a.strip()])
        assert n_args == len(_hip.SIGNATURES[name]), f"{name}: header has {n_args} args, ctypes table {len(_hip.SIGNATURES[name])}"
    # argument types, position by position
    import ctypes
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_int64: "int64"}
    for name, args in declared:
        want = []
        for a in [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]:
            if "*" in a:
                want.append("ptr")
            elif re.match(r"(const\s+)?int64_t\b", a):
                want.append("int64")
            elif re.match(r"(const\s+)?float\b", a):
                want.append("float")
            elif re.match(r"(const\s+)?int\b", a):
                want.append("int")
            else:
                raise AssertionError(f"{name}: unparsed argument {a!r}")
        got = [kinds[t] for t in _hip.SIGNATURES[name]]
        assert got == want, f"{name}: ctypes {got} != header {want}"
    assert L.ed_version() == _hip.ABI_VERSION
    assert L.ed_error_string(0)


def test_product_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from elasticdiffusion_official_amd import ElasticDiffusion, ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ElasticDiffusion(torch.device("cpu"), "1.5")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cfg_ddim_step(*(torch.zeros(8) for _ in range(5)), 1.0, 1.0, 1.0, 1.0, 1.0)


def test_package_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; import elasticdiffusion_official_amd as p; from elasticdiffusion_official_amd import pipeline, ops, "
            "geometry, host_rng, schedule, sharding; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=_hip.ROOT_DIR, timeout=300)


def test_every_kernel_defined_is_launched():
    """Static check of the .hip sources: every ``__global__`` kernel is referenced from live host code -- directly or through a
    launch macro that is actually expanded somewhere.  (A dispatch macro that was defined but never expanded once left an
    entry point returning success without launching anything; a container without a GPU cannot notice that at run time.)"""
    dead = []
    for src in _hip.SOURCES:
        text = re.sub(r"\\\n", " ", open(src).read())                                     # join macro continuation lines
        for m in re.finditer(r"^#define\s+(\w+)\(([^)]*)\)(.*)$", text, flags=re.M):
            name, body = m.group(1), m.group(3)
            rest = text[:m.start()] + text[m.end():]
            if ("<<<" in body or "LAUNCH" in body) and not re.search(r"\b%s\s*\(" % name, rest):
                dead.append(name)
                text = rest                                                                   # its body launches nothing
        defined = set(re.findall(r"__global__[^;{]*?\b(k_\w+)\s*\(", text, flags=re.S))
        assert defined, src
        for k in sorted(defined):
            uses = len(re.findall(r"\b%s\b" % k, text))
            assert uses >= 2, f"{k} in {src} is defined but never launched"
    assert not dead, f"launch macros never expanded: {dead}"
