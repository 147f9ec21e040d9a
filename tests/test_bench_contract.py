"""bench.py's contract pieces that do not need a GPU: the flag defaults the driver relies on, the committed PMC summary the
``roofline.traffic`` field is read from, and the loud failure when no MI355X is visible (there is no CPU fallback to time)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('vited_bench', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flag_defaults(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0 and a.workload == 'A-train'
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '7', '--warmup', '3'])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 3)


def test_committed_pmc_summary_covers_the_dominant_kernel_classes():
    bench = _bench()
    for cls in ('gemm_nt_mfma_kernel', 'gemm_tn_wide_kernel', 'gemm_row_kernel'):
        t = bench.pmc_traffic(cls)
        assert t['traffic'] and t['traffic'] > 1e6, cls          # bytes per launch
    assert bench.pmc_traffic('no_such_kernel') == {'traffic': None}
    assert bench.PEAK_BF16_DENSE_TFLOPS == 2500.0 and bench.PEAK_HBM_GBPS == 8000.0   # MI355X_MICROARCH.md sheet values


def test_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('a GPU is visible')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert 'no GPU is visible' in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout           # no JSON line from a machine that did not measure
