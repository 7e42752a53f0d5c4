"""The driver's build check: __graft_entry__.build() compiles / loads the library and imports the package on CPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_build_entry_point_runs_without_gpu():
    import __graft_entry__ as g
    g.build()  # up-to-date objects are reused, so this is seconds; asserts the ABI version against the header


def test_header_and_binding_agree_on_abi_version():
    import re
    from unispeech_amd import _lib
    src = open(os.path.join(ROOT, "include", "wavlm_hip.h")).read()
    assert int(re.search(r"#define WAVLM_HIP_ABI_VERSION (\d+)", src).group(1)) == _lib.ABI_VERSION
