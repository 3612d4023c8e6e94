"""No-GPU check of the drop-in boundary: libt2v_hip.so loads and exports every symbol that
include/t2v_hip.h declares, and the ctypes binding covers exactly that set."""
import ctypes
import os
import re

from t2v_turbo_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "t2v_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t2v_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert "t2v_gemm" in names and "t2v_attn_spatial" in names and len(names) >= 20
    assert os.path.exists(native.LIB_PATH), "build the library first: python __graft_entry__.py"
    lib = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/t2v_hip.h but not exported"
    assert sorted(native.EXPORTED) == names, set(native.EXPORTED) ^ set(names)


def test_struct_layout_matches_header():
    # field order of t2v_gemm_desc in the header == ctypes Structure
    text = open(os.path.join(ROOT, "include", "t2v_hip.h")).read()
    body = text[text.index("typedef struct t2v_gemm_desc {"):text.index("} t2v_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split(";")[:-1]:
        stmt = stmt.split("{")[-1].strip()
        if not stmt:
            continue
        names = re.sub(r"^(const\s+)?(void|float|int|unsigned|long long)\s*\*?", "", stmt)
        fields += [n.strip().lstrip("*") for n in names.split(",")]
    assert fields == [f[0] for f in native.GemmDesc._fields_], fields


def test_calls_without_gpu_fail_cleanly():
    lib = native.load()
    assert lib.t2v_version() >= 100
    assert lib.t2v_gemm(None, None) == -1  # T2V_EINVAL, no crash
    assert b"null" in lib.t2v_last_error()


def test_header_is_plain_c():
    """include/t2v_hip.h is the FFI contract: it must compile as C99 with nothing but the system headers."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "t2v_hip.h")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_tune_table_nearest_shape_fallback():
    """native.TuneTable: exact key first; else the entry with the same (mode, N, K, batch) and the nearest M within 0.3 .. 2.5x — its
    tile, and its K split only when M is within a third; anything else is a miss (library heuristic)."""
    from t2v_turbo_amd.native import TuneTable
    t = TuneTable({(0, 40960, 320, 320, 1): (23, 1), (0, 10240, 320, 320, 1): (11, 1), (1, 640, 1280, 11520, 1): (4, 4)}, nearest=True)
    assert t.lookup((0, 40960, 320, 320, 1)) == (23, 1)
    assert t.lookup((0, 30000, 320, 320, 1)) == (23, 0)       # nearest in ratio is 40960 (1.37x: tile only)
    assert t.lookup((0, 12288, 320, 320, 1)) == (11, 1)       # 10240 is within a third: split kept
    assert t.lookup((0, 8192, 320, 320, 1)) == (11, 1)
    assert t.lookup((0, 2048, 320, 320, 1)) is None           # 5x away from every entry
    assert t.lookup((0, 40960, 320, 640, 1)) is None          # another K
    assert t.lookup((1, 512, 1280, 11520, 1)) == (4, 4)
    assert t.lookup((1, 1024, 1280, 11520, 1)) == (4, 0)
    assert t.stats == {"exact": 1, "nearest": 5, "miss": 2}
    t.nearest = False
    assert t.lookup((0, 8192, 320, 320, 1)) is None
    assert TuneTable({}).lookup((0, 1, 1, 64, 1)) is None
