"""CPU: libdance_hip.so loads without a GPU and exports every symbol include/dance_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dance_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dance_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from dance_amd import _lib

    names = _declared()
    assert len(names) >= 15 and "dance_snake_evolve" in names
    h = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(h, n), "libdance_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes signature for %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().dance_abi_version() == _lib.ABI_VERSION


def test_no_compute_entry_point_needs_a_gpu_to_validate_arguments():
    """argument errors are reported before anything is enqueued (works on a CPU-only box)."""
    from dance_amd import _lib

    h = _lib.lib()
    rc = h.dance_uniform_upsample(9, None, 1, 4, 128, None, None, None, None)
    assert rc == 1 and b"requirement failed" in h.dance_last_error()
    rc = h.dance_circconv(None, 0, None, None, None, None, None, None, None, 0, 1, 130, 128, 128, 9, 1, 0, None)
    assert rc == 1            # v % 4 != 0
    assert h.dance_snake_workspace_bytes(0, 128) == 0
    assert h.dance_snake_workspace_bytes(100, 128) > 100 * 1024 * 128 * 4
    with pytest.raises(RuntimeError, match="requirement failed"):
        _lib.check(rc, "circconv")


def test_struct_layout_matches_header():
    from dance_amd import _lib

    # 4 + 28 + 2 + 1 + 2 + 2 + 2 pointers, 7 ints dilation, 1 int feat_channels
    assert ctypes.sizeof(_lib.SnakeNetStruct) == 41 * 8 + 8 * 4


def test_device_code_has_no_packed_fp32_forms():
    """The library is built with -packed-fp32-ops after v_pk_{fma,mul,add}_f32 gave wrong values in
    lanes 48..63 beside MFMA waves (dance_amd/build.py NOPK).  A toolchain that stops
    honouring the switch would bring them back silently: look at the ISA of every gfx950 code object
    of the library that ships."""
    from dance_amd import _lib, build

    if not os.path.exists(build.OBJDUMP):
        pytest.skip("llvm-objdump not installed")
    counts = build.device_isa_counts(_lib.SO_PATH)
    assert counts["code_objects"] >= len(build.SOURCES)
    assert counts["v_mfma_f32"] > 1000            # the disassembly really is the kernels'
    for name in build.PACKED_FP32:
        assert counts[name] == 0, "%d %s instructions in libdance_hip.so" % (counts[name], name)
