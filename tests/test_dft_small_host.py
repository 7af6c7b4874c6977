"""csrc/mvs_dft_small.h on the HOST: its functions are ``__host__ __device__``, so the whole-line transforms the GPU runs for
short composite axes (prime-factor / Cooley-Tukey split, dense symmetric prime leaves, compile-time twiddles) are compiled
for the CPU and compared with a direct double-precision DFT for every supported length up to 64.  Needs hipcc only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_supported_length_against_a_direct_dft(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    exe = tmp_path / "dft_small_host_test"
    cmd = [hipcc, "-O1", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "multiview-stitcher_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "dft_small_host_test.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = r.stdout.strip().splitlines()
    lengths = [int(ln.split()[0]) for ln in lines[:-1]]
    assert 51 in lengths and 49 in lengths and 27 in lengths and 57 in lengths and len(lengths) >= 45      # 3 x 17, 7 x 7, 3 x 9, 3 x 19, ...
    assert float(lines[-1].split()[1]) < 1e-6
