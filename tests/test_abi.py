"""CPU: the C-ABI shared library loads and exports every symbol include/vited.h declares.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built(vited):
    if not os.path.exists(vited._lib.LIB_PATH):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'vit-ed_amd', 'csrc'), '-j', '8'], check=True)


def test_header_binding_and_library_agree(vited):
    _ensure_built(vited)
    declared = vited._lib.header_declared_functions()
    assert declared == sorted(vited._lib.SIGNATURES), 'include/vited.h and the ctypes SIGNATURES table differ'
    lib = ctypes.CDLL(vited._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} is declared in include/vited.h but not exported'
    out = subprocess.run(['nm', '-D', '--defined-only', vited._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if ' T vited_' in l)
    assert exported == declared, 'exported vited_* symbols differ from the header'


def test_library_loads_and_reports_errors(vited):
    _ensure_built(vited)
    lib = vited._lib.load()
    assert lib.vited_abi_version() == 1
    assert lib.vited_strerror(0) == b'ok' and b'workspace' in lib.vited_strerror(4) and b'unknown' in lib.vited_strerror(99)
    # argument validation happens before any launch, so these are safe without a GPU
    assert lib.vited_gemm(None, 0, None, 0, 0, 1, 0, 0, 0, 0, None, None, None, None, None, 0, 0, 0, 0, 0, None) == 1
    assert lib.vited_layernorm_fwd(None, 0, None, None, None, 1, 0, None, None, 0, 0, 1e-6, None) == 1
    assert lib.vited_linear_bwd_weight_workspace_bytes(65536, 1152, 384) > 0
    assert lib.vited_last_gemm_path() == 0
