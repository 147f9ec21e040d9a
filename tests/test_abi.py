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


def test_row_complete_kernel_tile_heights(vited):
    """gemm_row.hip picks the tile height (96 / 128 / 144 / 160 rows) with the smallest makespan over rounds of 256 workgroups;
    the partial-rows query (= number of tiles, one [2][384] partial each) shows the choice without a GPU."""
    _ensure_built(vited)
    lib = vited._lib.load()
    rows = lib.vited_linear_layernorm_bwd_partial_rows
    assert rows(65536) == 512          # 64-token batches of 1024: two full rounds of 128-row tiles
    assert rows(66560) == 463          # 65-token batches: 144-row tiles (two rounds; 520 x 128 would need three)
    assert rows(72 * 1025) == 462      # config H decoder, 72 pairs: 160-row tiles (two rounds; 577 x 128 would need three)
    assert rows(24 * 1024) == 256      # config H encoder, 24 images: one round of 96-row tiles
    assert rows(100) == 2              # 96 + 4 rows
    assert lib.vited_linear_layernorm_supported(4096, 384, 384) == 1
    assert lib.vited_linear_layernorm_supported(4096, 384, 96) == 0      # the contraction comes in K = 64 tiles
    assert lib.vited_linear_layernorm_supported(4096, 512, 384) == 0     # rows are 384 wide
    assert lib.vited_linear_layernorm_bwd_workspace_bytes(65536, 384) == 512 * 2 * 384 * 4
