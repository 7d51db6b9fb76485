"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/unetmi.h declares."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "unetmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(umi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from umi import build, lib
    names = _declared()
    assert len(names) >= 15
    cdll = ctypes.CDLL(build.LIB)
    missing = [n for n in names if not hasattr(cdll, n)]
    assert not missing, f"declared in unetmi.h but not exported: {missing}"
    assert set(lib.SIGNATURES) == set(names), set(lib.SIGNATURES) ^ set(names)
    assert lib.fn("umi_arch")() == b"gfx950"
    assert lib.fn("umi_version")() >= 1


def test_bad_arguments_return_status_not_crash():
    from umi import lib
    # null pointers / non-positive sizes are rejected before any launch (no GPU touched)
    assert lib.fn("umi_bn_finalize")(None, 0, 0, 0.0, None, None, 1e-5, 0.1, None, None, None, None, None) == -1
    assert lib.fn("umi_conv_fwd")(None, 0, None, None, None, None, 0, None, 1, 1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 0, 0, 1, 1,
                                  0, 0, 0, None) == -1
    assert lib.fn("umi_pool2_fwd")(None, 0, None, None, 0, 0, 0, 0, 0, 0, None) == -1
