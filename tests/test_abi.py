"""The C-ABI library builds for gfx950, loads, and exports every symbol include/jmhip.h declares (no GPU)."""
import ctypes


def test_library_exports_declared_symbols(pkg):
    pkg.build_library()
    lib = pkg.load_library()
    names = pkg.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.jmhip_abi_version() == 1


def test_fails_loudly_without_gpu_or_bad_args(pkg):
    import torch
    lib = pkg.load_library()
    # bad geometry is rejected before any device work
    try:
        pkg.Context(100, 50)
        assert False, "expected an error"
    except pkg.JmhipError:
        pass
    if not torch.cuda.is_available():
        try:
            pkg.Context(64, 48)
            assert False, "no GPU: context creation must fail, not fall back"
        except pkg.JmhipError as e:
            assert "HIP" in str(e) or "device" in str(e)
    assert lib.jmhip_strerror(3) != ctypes.c_char_p(None)
