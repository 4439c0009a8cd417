"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/geosplat_hip.h
declares (and nothing is declared that is not exported), and the product path fails LOUDLY without a GPU /
without the library -- it never falls back to the oracle or to PyTorch."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "geosplat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from geosplatting_amd import _lib
    lib = _lib.lib()
    decl = _declared()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in geosplat_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == decl
    assert lib.gs_version() >= 100
    assert lib.gs_last_error() is not None


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device (no compute, no GPU needed)."""
    from geosplatting_amd import _lib
    lib = _lib.lib()
    assert lib.gs_project_ws_bytes(0) > 0 and lib.gs_project_ws_bytes(1 << 21) >= 4 * 8 * (1 << 11)
    assert lib.gs_raster_ws_bytes(ctypes.c_int64(1000), 500, 800, 800, 16) >= 3 * 16 * 1000 + 4 * 2500 + 3 * 16 * 500
    rc = lib.gs_raster_fwd(0, 0, 16, 3, 0, None, None, None, None, None, ctypes.c_int64(0), None, None, None, None, None,
                           None, ctypes.c_size_t(0), None)
    assert rc == -1 and b"bad image size" in lib.gs_last_error()
    rc = lib.gs_raster_fwd(16, 16, 8, 3, 0, None, None, None, None, None, ctypes.c_int64(0), None, None, None, None, None,
                           None, ctypes.c_size_t(0), None)
    assert rc == -1 and b"tile_size" in lib.gs_last_error()
    rc = lib.gs_shade_fwd(1, None, None, None, None, None, ctypes.c_float(0.1), ctypes.c_float(1.0), 7, None, None, None)
    assert rc == -1
    # FlexiCubes / field entry points
    assert lib.gs_flexicubes_ws_bytes(0, 4, 4) == 0 and lib.gs_flexicubes_ws_bytes(4, 4, 4) > 4 * 64 + 2 * 4 * 64 + 4 * 3 * 125
    assert lib.gs_flexicubes_ws_bytes(2000, 2000, 2000) == 0                       # 3 * vertices would overflow int32
    rc = lib.gs_flexicubes_count(0, 4, 4, None, None, ctypes.c_size_t(0), None, None)
    assert rc == -1 and b"resolution" in lib.gs_last_error()
    rc = lib.gs_flexicubes_count(4, 4, 4, None, None, ctypes.c_size_t(8), None, None)
    assert rc == -1 and b"workspace" in lib.gs_last_error()
    rc = lib.gs_mlp_wgrad(ctypes.c_int64(10), 33, 32, None, None, ctypes.c_float(1.0), None, 0, None, ctypes.c_size_t(0), None)
    assert rc == -1 and b"[1, 32]" in lib.gs_last_error()
    lib.gs_mlp_wgrad_ws_bytes.restype = ctypes.c_size_t
    assert lib.gs_mlp_wgrad_ws_bytes(ctypes.c_int64(1 << 21)) == 1024 * 1024 * 4
    rc = lib.gs_hashgrid_fwd(10, 16, 3, 18, None, None, None, None, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    import geosplatting_amd as gs
    from geosplatting_amd._lib import GeoSplatHipError
    z = torch.zeros(1, 3)
    with pytest.raises(GeoSplatHipError, match="GPU only"):
        gs.rasterization(z, torch.ones(1, 4), z + 1, torch.ones(1), z, torch.eye(4)[None], torch.eye(3)[None], 16, 16)
    with pytest.raises(GeoSplatHipError):
        gs.as_splitsum(torch.rand(6, 64, 64, 3))
    with pytest.raises(GeoSplatHipError):
        gs.tone_map(torch.rand(4, 4, 4), torch.tensor(1.0))


def test_missing_library_fails_loudly(tmp_path):
    """With the .so absent the import of an op must raise -- no silent CPU path."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import geosplatting_amd._lib as L\n"
        "L.LIB_PATH = %r\n"
        "try:\n"
        "    L.lib()\n"
        "except L.GeoSplatHipError as e:\n"
        "    print('RAISED', 'no CPU' in str(e))\n" % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "geosplatting_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), fn
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py may use the oracle only inside its cpu_baseline leg (functions named cpu_baseline*)
    parts = bench.split("\ndef ")
    legs = [q for q in parts if q.startswith("cpu_baseline")]
    assert legs and all("import oracle" in q for q in legs)
    assert all("import oracle" not in q for q in parts if not q.startswith("cpu_baseline"))


def test_prefetch_registers_are_never_copied(tmp_path):
    """The compositor kernels prefetch three raw batches with inline-assembly loads and wait for them with a static
    `s_waitcnt vmcnt(6)` (csrc/gs_raster.hip: raw_load / raw_wait / raw_drain).  The compiler does not know that those registers
    are in flight between the load and the wait: if its register allocator ever split such a live range (a `v_mov`, an AGPR or
    scratch spill of the destination registers), the copy would read data that has not arrived.  This test disassembles the
    shipped kernels and checks that, after the initial loads, no move / spill instruction has one of the prefetch registers as
    its SOURCE -- the build is only valid while that holds (a register-pressure change in these kernels can break it)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from geosplatting_amd import build as B
    out = tmp_path / "raster.s"
    subprocess.check_call([hipcc, *[f for f in B.FLAGS if f not in ("-shared", "-fPIC")], "-S", "--cuda-device-only",
                           os.path.join(B.CSRC, "gs_raster.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    src = out.read_text().split("\n")

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]$", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    checked = 0
    for i, line in enumerate(src):
        if not re.match(r"_Z2[345]raster_(fwd_lanes|fwd_window|bwd_lanes|bwd_lanes2)_kernelILi\d+E", line):
            continue
        end = next(j for j in range(i, len(src)) if "s_endpgm" in src[j])
        # the prefetch loads are the INLINE-ASSEMBLY ones (";;#ASMSTART" in front of them): the compiler's own 16-byte loads (e.g. the
        # backward's checkpoint reads) are tracked by its wait-count pass and may live anywhere
        body, asm_line = [], set()
        for l in src[i:end]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                asm_line.add(len(body))
            if t and not t.startswith(";"):
                body.append(t)
        loads = [k for k, l in enumerate(body) if l.startswith("global_load_dwordx4") and k in asm_line]
        if len(loads) < 9:
            continue
        raw = set()
        for k in loads:
            raw |= regs(body[k].split()[1].strip(","))
        assert len(raw) == 36, (line, len(raw))                      # three batches x three 16-byte records, fixed registers
        drain = next(k for k in range(loads[-1], len(body)) if body[k].startswith("s_waitcnt vmcnt(0)"))   # raw_drain: all arrived
        for l in body[loads[8] + 1:drain]:
            if not re.match(r"(v_mov_b32|v_mov_b64|v_accvgpr_write|scratch_store|buffer_store)", l):
                continue
            ops = [t.strip(",") for t in l.split()[1:]]
            srcs = set().union(*[regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
            assert not (srcs & raw), f"{line.split(':')[0]}: prefetch register copied while possibly in flight: {l}"
        checked += 1
    assert checked >= 4
