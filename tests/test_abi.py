"""No-GPU check of the drop-in boundary: libt2v_hip.so loads and exports every symbol that
include/t2v_hip.h declares, and the ctypes binding covers exactly that set."""
import ctypes
import os
import re

from t2v_turbo_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(experimental=False):
    """Function names the header declares: the product view (the T2V_EXPERIMENTAL blocks dropped) or those blocks alone."""
    text = open(os.path.join(ROOT, "include", "t2v_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    blocks = re.findall(r"#ifdef T2V_EXPERIMENTAL(.*?)#endif", text, flags=re.S)
    text = "\n".join(blocks) if experimental else re.sub(r"#ifdef T2V_EXPERIMENTAL.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t2v_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert "t2v_gemm" in names and "t2v_attn_spatial" in names and len(names) >= 20
    assert os.path.exists(native.LIB_PATH), "build the library first: python __graft_entry__.py"
    lib = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/t2v_hip.h but not exported"
    assert sorted(native.EXPORTED) == names, set(native.EXPORTED) ^ set(names)
    # the measured negative results live behind T2V_EXPERIMENTAL: declared there, bound when present, NOT in the product library
    exp = _declared(experimental=True)
    assert exp == sorted(native.EXPERIMENTAL), set(exp) ^ set(native.EXPERIMENTAL)
    assert not any(hasattr(lib, n) for n in exp), "the product library exports T2V_EXPERIMENTAL entry points"


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


def test_split_hint_survives_a_nearest_shape_hit_without_a_trusted_split():
    """A tile-table miss used to keep the caller's split_k hint; the nearest-shape fallback returns (tile, 0) when the tuned M is not
    within a third of the asked one, and that 0 must not overwrite the hint (the training engine's token-contracted weight gradients
    pass one).  Descriptor construction only: no launch, no GPU."""
    import torch
    ops = native.HipOps()
    M_t, N, K = 1024, 128, 256
    ops.tune = native.TuneTable({(native.GEMM_LINEAR, M_t, N, K, 1): (6, 3)}, nearest=True)
    w = torch.zeros(N, K, dtype=torch.bfloat16)

    def desc(M, **kw):
        return ops._gemm_desc(torch.zeros(M, K, dtype=torch.bfloat16), w, torch.zeros(M, N, dtype=torch.bfloat16), M=M, N=N, **kw)

    d = desc(M_t, split_k=5)
    assert (d.tile_cfg, d.split_k) == (6, 3)            # exact hit: the tuned entry wins over the hint
    d = desc(1100, split_k=5)
    assert (d.tile_cfg, d.split_k) == (6, 3)            # nearest, M within a third: tile and split of the tuned entry
    d = desc(2200, split_k=5)
    assert (d.tile_cfg, d.split_k) == (6, 5)            # nearest tile, split not trusted: the caller's hint stands
    d = desc(2200)
    assert (d.tile_cfg, d.split_k) == (6, 0)            # ... and without a hint the library's own rule
    d = desc(8000, split_k=5)
    assert (d.tile_cfg, d.split_k) == (0, 5)            # miss: as before
