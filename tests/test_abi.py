"""The C-ABI library loads and exports every symbol include/rt_hip.h declares (no compute,
no GPU needed), and so does the oracle that mirrors it."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "rt_hip.h").read_text()
DECLARED = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", HEADER)))


def test_header_declares_the_binding():
    from raytracer_amd import ffi

    assert set(ffi.ABI_SYMBOLS) == set(DECLARED)
    assert len(DECLARED) >= 25


@pytest.mark.parametrize("so", ["raytracer_amd/librt_hip.so", "oracle/librt_oracle.so"])
def test_library_exports_all_symbols(so):
    path = ROOT / so
    if not path.exists():
        import __graft_entry__ as g

        g.build()
    dll = ctypes.CDLL(str(path))
    for name in DECLARED:
        assert hasattr(dll, name), f"{so} does not export {name}"
    dll.rt_backend_name.restype = ctypes.c_char_p
    assert dll.rt_backend_name() == (b"hip" if "hip" in so else b"oracle")


def test_product_has_no_cpu_fallback(tmp_path):
    from raytracer_amd import ffi

    with pytest.raises(FileNotFoundError):
        ffi.RtLib(tmp_path / "missing.so")
    # nothing under raytracer_amd/ loads, links or imports the oracle
    for f in (ROOT / "raytracer_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".h"):
            text = f.read_text()
            assert "librt_oracle" not in text and "np_ref" not in text and "import oracle" not in text, f


def test_struct_layouts_match_the_header(tmp_path):
    """rt_stats_t / rt_info_t as gcc lays them out from include/rt_hip.h == the ctypes structures of the binding, size and
    every field offset (a field added to one side only would shift everything after it)."""
    import subprocess

    from raytracer_amd import ffi

    fields = {"rt_stats_t": [k for k, _ in ffi.RtStats._fields_], "rt_info_t": [k for k, _ in ffi.RtInfo._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_hip.h"', "int main(void) {"]
    for name, fs in fields.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for f in fs:
            lines.append(f'  printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in (("rt_stats_t", ffi.RtStats), ("rt_info_t", ffi.RtInfo)):
        assert int(got[name]) == ctypes.sizeof(cls), name
        for f in fields[name]:
            assert int(got[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"
