"""CPU: the C-ABI library builds, loads and exports every symbol include/revo.h
declares (no compute calls: there is no GPU in this tier)."""
import os
import re

import reverso_amd  # noqa: F401
from reverso_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(experiments=False):
    text = open(os.path.join(ROOT, "include", "revo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    exp = "".join(re.findall(r"#ifdef REVO_EXPERIMENTS(.*?)#endif", text, flags=re.S))
    text = re.sub(r"#ifdef REVO_EXPERIMENTS.*?#endif", "", text, flags=re.S)
    names = lambda t: sorted(set(re.findall(r"\b(revo_[a-z0-9_]+)\s*\(", t)))
    return (names(exp), names(text))[0 if experiments else 1]


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"librevo.so does not export {n}"


def test_binding_covers_header_exactly():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_experiment_switches_are_not_in_the_product_library():
    """The work-skipping timing switches (wrong results) exist only in librevo_exp.so (`make exp`)."""
    lib = _lib.load()
    exp = _declared(experiments=True)
    assert "revo_op_set_gemm_debug" in exp and sorted(exp) == sorted(_lib.EXPERIMENT_SIGNATURES)
    for n in exp:
        assert not hasattr(lib, n), f"librevo.so exports the experiment switch {n}"
    # the parity-test hooks too: the product library cannot stop a forward early, hand out intermediate buffers, or be
    # put into a search mode that skips (or forces) the exact fallback
    for n in ("revo_vit_set_debug_layers", "revo_vit_read_residual", "revo_vit_read_tap", "revo_search_set_mode"):
        assert n in exp and not hasattr(lib, n), n
    xl = _lib.load_exp()
    for n in exp:
        assert hasattr(xl, n), f"librevo_exp.so does not export {n}"


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.revo_version() >= 100
    assert isinstance(lib.revo_last_error(), bytes)


def test_argument_validation_without_gpu():
    # these return before touching the device
    lib = _lib.load()
    assert lib.revo_gallery_clear(None) != 0
    assert b"null" in lib.revo_last_error()
    assert lib.revo_gallery_size(None) == -1
    assert lib.revo_vit_seq_len(None) == -1


def test_product_has_no_oracle_import():
    pkg = os.path.join(ROOT, "revers-o_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), encoding="utf-8").read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_host_paths_under_address_and_ub_sanitizers():
    """api.hip's host code (argument validation, the checkpoint name / size map, error strings, handle lifetime,
    graceful failure without a device) compiled with -fsanitize=address,undefined and driven through every entry
    point by tests/native/abi_host_check.cpp.  (GPU sanitizers are not available on this pool: CPU build only.)"""
    import subprocess
    csrc = os.path.join(ROOT, "revers-o_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-j", "4", "all", "asan"], check=True, capture_output=True, timeout=900)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(csrc, "build", "asan", "abi_host_check")], capture_output=True, text=True, env=env,
                       timeout=120)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
