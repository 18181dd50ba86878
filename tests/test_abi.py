"""CPU: the C-ABI library loads and exports exactly what include/m2s.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from mesh2splat_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "m2s.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(m2s_[a-z_0-9]+)\s*\(", src))


def test_header_and_binding_agree():
    assert header_functions() == set(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(hiplib):
    for name in sorted(header_functions()):
        assert hasattr(hiplib, name), f"{name} declared in include/m2s.h but not exported by libm2s_hip.so"
    assert hiplib.m2s_abi_version() == 1


def test_struct_layouts_match_header():
    # m2s_gaussian == utils::GaussianDataSSBO: 6 x vec4 = 96 bytes; m2s_mesh: pointers + PODs
    assert C.sizeof(_lib.Texture) == 16
    assert C.sizeof(_lib.MeshC) == 8 + 4 + 4 + 12 + 12 + 16 + 3 * 16
    from mesh2splat_amd.scene import RECORD_FLOATS
    assert RECORD_FLOATS * 4 == 96


def test_prepass_struct_layout_matches_header(tmp_path):
    """m2s_prepass_params / m2s_quad as the C compiler lays them out == the ctypes mirror in mesh2splat_amd/prepass.py."""
    import shutil
    import subprocess
    from mesh2splat_amd.prepass import PrepassParamsC
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    fields = [f[0] for f in PrepassParamsC._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "m2s.h"\nint main(void){\n'
                   'printf("%zu %zu\\n", sizeof(m2s_prepass_params), sizeof(m2s_quad));\n' +
                   "".join('printf("%%zu\\n", offsetof(m2s_prepass_params, %s));\n' % f for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == C.sizeof(PrepassParamsC) and int(out[1]) == 96
    assert [int(v) for v in out[2:]] == [getattr(PrepassParamsC, f).offset for f in fields]


def test_no_cpu_fallback(hiplib):
    """Without a usable HIP device the product fails loudly instead of computing on the CPU."""
    h = C.c_void_p()
    st = hiplib.m2s_create(0, C.byref(h))
    if st == 0:          # a GPU is present (this test also runs on the GPU box)
        hiplib.m2s_destroy(h)
        return
    assert st == 2       # M2S_ERR_NO_DEVICE
    assert b"no CPU path" in hiplib.m2s_last_error(None)
    assert hiplib.m2s_create(-1, C.byref(h)) != 0
    assert hiplib.m2s_create(0, None) == 1


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under mesh2splat_amd/ or include/ may reference it."""
    bad = []
    for base in ("mesh2splat_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "_build" in dp:
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|m2s_oracle|orc_", txt) and f != "_lib.py":
                        bad.append(os.path.join(dp, f))
                    if f == "_lib.py" and re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_gfx950_has_no_image_sampling(tmp_path):
    """the north star's 'bindless image sampling' cannot exist on MI355X — hipcc refuses tex2DLod for gfx950.
    (Keeps that statement honest: if a future toolchain accepts it, this test fails and the design should be revisited.)"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "tools", "probe_image_support.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "--offload-device-only", "-S", src, "-o", str(tmp_path / "x.s")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "image/texture API not supported" in r.stderr


def test_integration_md_quotes_the_compiled_dropin():
    """INTEGRATION.md's reference-side replacement is the file oracle/Makefile compiles into oracle/_ref/ref_dropin_check."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = open(os.path.join(root, "oracle", "ref_dropin", "ConversionPassHip.cpp")).read()
    assert body in open(os.path.join(root, "INTEGRATION.md")).read()


def test_rccl_stand_in_exports_what_load_rccl_resolves():
    """tests/stub_rccl (the multi-process tests' RCCL stand-in) must export every symbol csrc/m2s_dist.cpp dlsym()s."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tests", "stub_rccl", "_build", "librccl_stub.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(so))], check=True, stdout=subprocess.DEVNULL)
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', open(os.path.join(root, "mesh2splat_amd", "csrc", "m2s_dist.cpp")).read()))
    assert len(wanted) == 9
    have = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert all(re.search(rf"\bT {w}\b", have) for w in wanted), wanted


def test_unloadable_rccl_path_is_an_error_string_not_a_crash(hiplib):
    """with M2S_RCCL_PATH naming a file that cannot be loaded, m2s_dist_unique_id must fail with a message
    (it used to build the message from a second dlerror() call, which returns NULL: undefined behaviour, a segfault in practice).
    No GPU needed: the library is loaded, librccl is not."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "from mesh2splat_amd import _lib\n"
        "L = _lib.load()\n"
        "buf = (ctypes.c_uint8 * 128)()\n"
        "L.m2s_dist_unique_id.restype = ctypes.c_int\n"
        "rc = L.m2s_dist_unique_id(buf)\n"
        "L.m2s_dist_last_error.restype = ctypes.c_char_p\n"
        "L.m2s_dist_last_error.argtypes = [ctypes.c_void_p]\n"
        "msg = L.m2s_dist_last_error(None).decode()\n"
        "print(rc, '|', msg)\n"
    )
    env = dict(os.environ, M2S_RCCL_PATH="/nonexistent/librccl_nowhere.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    rc, msg = r.stdout.strip().splitlines()[-1].split(" | ", 1)
    assert int(rc) != 0 and "M2S_RCCL_PATH" in msg and "librccl_nowhere" in msg, r.stdout
