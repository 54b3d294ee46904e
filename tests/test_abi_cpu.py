"""CPU: the C-ABI library loads, exports every symbol include/pyrad_hip.h declares, and
fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "pyrad_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int)\s+(lbl_[a-z_0-9]+)\s*\(", text, flags=re.M)
    assert len(names) >= 35
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound():
    from pyrad_amd import _native
    lib = _native.load()
    names = declared_symbols()
    for name in names:
        assert hasattr(lib, name), "libpyrad_hip.so does not export %s" % name
    # the ctypes binding declares a signature for each of them and nothing else
    assert sorted(_native.SIGNATURES) == names
    assert lib.lbl_abi_version() == 5


def test_struct_layouts_match_header():
    from pyrad_amd import _native
    assert ctypes.sizeof(_native.IsoParams) == 6 * 8
    assert ctypes.sizeof(_native.Grid) == 4 * 8 + 5 * 8
    assert _native.Grid.shard_first.offset == 56 and _native.Grid.window.offset == 48


def test_no_gpu_is_loud_not_a_fallback():
    from pyrad_amd import _native, engine, model, data, synthetic
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.NoDeviceError) as e:
        _native.Context(0)
    assert e.value.code == -2
    # the object model reaches the same wall at the first computation
    engine.shutdown()
    data.set_source(data.synthetic_source({"co2": synthetic.make_lines(1, 8, 595, 705)}))
    model.Layer.hasAtmosphere = False
    layer = model.Layer(10, 296, 1013.25, 600, 700)
    with pytest.raises(_native.NoDeviceError):
        layer.addMolecule("co2", ppm=400)          # getData -> createLineSurvey needs the device
    data.set_source(None)


def test_product_code_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(REPO, "pyrad_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "pyrad_oracle" not in src and "c_oracle" not in src, f
    bench = open(os.path.join(REPO, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench)]
    body_cpu = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    body_chk = bench[bench.index("def oracle_check"):]
    assert len(uses) == body_cpu.count("from oracle import") + body_chk.count("from oracle import")


def test_header_is_valid_c99_and_a_plain_c_host_links(tmp_path):
    """include/pyrad_hip.h is a C header (the boundary has no C++ in it): examples/abi_smoke.c compiles with
    gcc -std=c99 -pedantic, links against the shared library, and - without a GPU - reports the missing device
    through the status code instead of computing anything."""
    import subprocess
    src = os.path.join(REPO, "examples", "abi_smoke.c")
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.join(REPO, "pyrad_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(REPO, "include"),
                           src, "-L", lib_dir, "-lpyrad_hip", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe])
    from pyrad_amd import _native
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if _native.device_count() > 0:
        assert p.returncode == 0, p.stdout + p.stderr
        assert "rel err" in p.stdout and "0/0/1" in p.stdout and "[4502..5498]" in p.stdout
    else:
        assert p.returncode == 77 and "no HIP device" in p.stderr
