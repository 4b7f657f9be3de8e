"""The C-ABI library: builds for gfx950 without a GPU, loads, exports every symbol the header
declares, its structs have the layout the ctypes binding assumes, and argument validation
(no compute, no device) returns the documented error codes."""
import ctypes as C
import os
import re
import subprocess

import pytest

from hpmn_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hpmn_hip.h")


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.load()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hpmn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound(lib):
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "missing export " + n
        assert n in _lib.SIGNATURES, "binding has no signature for " + n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.hpmn_abi_version() == _lib.HPMN_ABI_VERSION


def test_struct_layout_matches_c_compiler(tmp_path):
    """sizeof/offsetof from gcc on the real header vs the ctypes mirror."""
    structs = {"HpmnInputProj": _lib.HpmnInputProj, "HpmnGruFwd": _lib.HpmnGruFwd,
               "HpmnGruBwd": _lib.HpmnGruBwd, "HpmnGruWgrad": _lib.HpmnGruWgrad, "HpmnReadDesc": _lib.HpmnReadDesc,
               "HpmnScanDesc": _lib.HpmnScanDesc, "HpmnOnlineUpdate": _lib.HpmnOnlineUpdate,
               "HpmnGruFusedFwd": _lib.HpmnGruFusedFwd, "HpmnGruPairFwd": _lib.HpmnGruPairFwd, "HpmnGruPairBwd": _lib.HpmnGruPairBwd, "HpmnPipe": _lib.HpmnPipe,
               "HpmnTrainLayout": _lib.HpmnTrainLayout, "HpmnScatterPlan": _lib.HpmnScatterPlan,
               "HpmnRowsAdam": _lib.HpmnRowsAdam, "HpmnTileFwd": _lib.HpmnTileFwd, "HpmnTrainStep": _lib.HpmnTrainStep}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hpmn_hip.h"', "int main(void){"]
    for name, st in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for fname, _ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, fname, name, fname))
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for name, st in structs.items():
        assert int(got[name]) == C.sizeof(st), name
        for fname, _ in st._fields_:
            assert int(got["%s.%s" % (name, fname)]) == getattr(st, fname).offset, (name, fname)


def test_error_codes_without_touching_a_device(lib):
    assert lib.hpmn_strerror(0) == b"ok"
    assert b"invalid" in lib.hpmn_strerror(-1)
    assert lib.hpmn_gru_shape_supported(32, 48) == 1 and lib.hpmn_gru_shape_supported(64, 64) == 1
    assert lib.hpmn_gru_shape_supported(128, 32) == 1 and lib.hpmn_gru_shape_supported(128, 128) == 1
    assert lib.hpmn_gru_shape_supported(128, 48) == 0 and lib.hpmn_gru_shape_supported(96, 32) == 0
    assert lib.hpmn_gru_shape_supported(64, 40) == 0
    # null struct / null pointers -> EINVAL before any launch
    assert lib.hpmn_gru_scan_fwd(None, None) == -1
    assert lib.hpmn_gru_input_proj(None, None) == -1
    a = _lib.HpmnGruFwd()
    a.B, a.T, a.D, a.H = 4, 10, 32, 64
    assert lib.hpmn_gru_scan_fwd(C.byref(a), None) == -1
    assert lib.hpmn_adam_step(None, None, None, None, 16, 0.1, 0.9, 0.999, 1e-8, 1.0, 1.0, None) == -1
    assert lib.hpmn_adam_step(None, None, None, None, 0, 0.1, 0.9, 0.999, 1e-8, 1.0, 1.0, None) == 0   # n == 0
    assert lib.hpmn_embed_gather(None, None, None, 0, 3, 16, 10, 1, None) == 0                          # N == 0
    assert lib.hpmn_embed_gather(None, None, None, 5, 3, 6, 10, 1, None) == -2                          # E % 4
    assert lib.hpmn_embed_grad_scatter(None, None, None, 2, 5, 3, 24, 0, 10, 1, None) == -2             # 64 % E
    assert lib.hpmn_table_mark_rows(None, 0, None, 10, 2, None) == 0 and lib.hpmn_table_mark_rows(None, 4, None, 10, 2, None) == -1
    t128 = _lib.HpmnTileFwd()
    assert lib.hpmn_tile_fwd(None, None) == -1 and lib.hpmn_tile_supported(128, 32) == 1 and lib.hpmn_tile_supported(64, 48) == 1
    assert lib.hpmn_tile_supported(32, 32) == 0
    t128.B, t128.T, t128.D, t128.H, t128.period = 4, 8, 48, 128, 2
    assert lib.hpmn_tile_fwd(C.byref(t128), None) == -2                                                     # H = 128: D not 32 / 128
    t128.D = 128
    assert lib.hpmn_tile_fwd(C.byref(t128), None) == -1                                                     # neither x nor xp
    ra = _lib.HpmnRowsAdam()
    assert lib.hpmn_rows_sum_adam(None, None) == -1 and lib.hpmn_rows_sum_adam(C.byref(ra), None) == -1     # world == 0
    ra.world, ra.E, ra.V = 2, 16, 100
    assert lib.hpmn_rows_sum_adam(C.byref(ra), None) == 0                                                   # empty windows
    ra.n[1], ra.rows_stride, ra.ids_stride = 5, 8, 8
    assert lib.hpmn_rows_sum_adam(C.byref(ra), None) == -1                                                  # null buffers
    ra.E = 24
    assert lib.hpmn_rows_sum_adam(C.byref(ra), None) == -2                                                  # E/4 not a power of two
    ra.E, ra.world = 16, 9
    assert lib.hpmn_rows_sum_adam(C.byref(ra), None) == -1                                                  # > HPMN_MAX_RANKS
    assert lib.hpmn_table_mark_ranks(None, 0, 2, None, 0, 0, None, 10, 0, None, 0, 0, None) == 0
    assert lib.hpmn_table_mark_ranks(None, 8, 2, None, 0, 8, None, 10, 0, None, 0, 0, None) == -1
    plan = _lib.HpmnScatterPlan()
    plan.n = 30
    assert lib.hpmn_embed_grad_segsum(C.byref(plan), None, None, 2, 5, 3, 24, 0, 0, None, 0, None) == -2  # 256 % (E/4)
    assert lib.hpmn_embed_grad_segsum(C.byref(plan), None, None, 2, 5, 4, 16, 0, 0, None, 0, None) == -1  # n != B*T*F
    assert lib.hpmn_embed_grad_segsum(C.byref(plan), None, None, 2, 5, 3, 16, 0, 0, None, 0, None) == -1  # null arrays
    assert lib.hpmn_embed_grad_segsum_partials_floats(64, 16) >= 2 * (64 // lib.hpmn_embed_grad_segsum_chunk()) * 16 and lib.hpmn_scatter_plan(None, 0, 0, None, None, None, None, None) == 0
    # workspace size / divisibility of the layer lengths (the tf.reshape at code/hpmn.py:124)
    d = _lib.HpmnScanDesc()
    d.B, d.T, d.F, d.E, d.H, d.K, d.V = 128, 100, 3, 16, 32, 3, 1000
    d.front_zero, d.mask_id0, d.last_index = 0, 1, -1
    for i, p in enumerate((2, 2, 5)):
        d.periods[i] = p
    need = lib.hpmn_scan_workspace_bytes(C.byref(d))
    assert need >= 128 * 100 * 96 * 4 + 2 * 128 * 50 * 32 * 4
    d.periods[0] = 3
    assert lib.hpmn_scan_workspace_bytes(C.byref(d)) == 0


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.HpmnLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under hpmn_amd/ may reference it."""
    pkg = os.path.join(ROOT, "hpmn_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn
                assert "oracle." not in txt and "oracle/" not in txt, fn
