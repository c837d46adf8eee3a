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
    lib = pkg.load_library()
    # bad geometry is rejected before any device work
    try:
        pkg.Context(100, 50)
        assert False, "expected an error"
    except pkg.JmhipError:
        pass
    # without a GPU, context creation must fail with a device error (never fall back to the CPU)
    try:
        ctx = pkg.Context(64, 48)
        ctx.close()          # a GPU is present
    except pkg.JmhipError as e:
        assert "HIP" in str(e) or "device" in str(e)
    assert lib.jmhip_strerror(3) is not None


def test_python_package_has_no_cpu_compute_path(pkg):
    """The binding only marshals: the product path must not import the oracle."""
    import os
    here = os.path.dirname(pkg.__file__)
    for fn in os.listdir(here):
        if fn.endswith(".py"):
            txt = open(os.path.join(here, fn)).read()
            assert "oracle" not in txt, fn
