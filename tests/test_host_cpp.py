"""The C++ host-side mirror of the reference API (scann_rust_amd/host/scann.hpp): compiles
on CPU; on a GPU its test program replays the reference's own searcher unit tests
(brute_force/searcher.rs:280-376, hashes/hasher.rs:322-380, tree_x_hybrid/mod.rs:436-468)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "scann_rust_amd", "host")


def _compile():
    from scann_rust_amd import build
    build.build()
    exe = os.path.join(HOST, "host_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe,
                           os.path.join(HOST, "host_test.cpp"), "-L" + os.path.join(ROOT, "scann_rust_amd"),
                           "-lscann_hip", "-Wl,-rpath," + os.path.join(ROOT, "scann_rust_amd")])
    return exe


def test_host_mirror_compiles_and_fails_loudly_without_gpu():
    exe = _compile()
    import torch
    if torch.cuda.device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stdout   # Unavailable, no CPU fallback


@pytest.mark.gpu
def test_host_mirror_reference_unit_tests():
    exe = _compile()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_test ok" in r.stdout
