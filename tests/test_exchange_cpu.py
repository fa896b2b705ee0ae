"""CPU (-m "not gpu"): the C++ host layer's exchange (host/se3d_exchange.{hpp,cpp}) builds with plain g++ — RCCL and the HIP runtime are bound
with dlopen at run time — and the rendezvous of its TEST-ONLY staged all-gather is correct: host/se3d_exchange_test runs `world` ranks as threads
over plain host memory, several calls of varying size, and checks every rank's gathered buffer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,calls", [(2, 5), (5, 9)])
def test_staged_all_gather_rendezvous(world, calls):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "host", "se3d_exchange_test"), str(world), str(calls)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "%d ranks x %d calls, 0 mismatching blocks" % (world, calls) in r.stdout


def test_host_library_exports_the_exchange():
    """libse3d_host.so carries the exchange classes and needs neither librccl nor a HIP header to build."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    syms = subprocess.run(["nm", "-DC", os.path.join(ROOT, "host", "libse3d_host.so")], capture_output=True, text=True, check=True).stdout
    for name in ("se3d::make_rccl_collective", "se3d::make_staged_collective", "se3d::HypothesisShard::run_stage1", "se3d::HypothesisShard::run_stage2",
                 "se3d::SkeletonExchange::exchange"):
        assert name in syms, name
    deps = subprocess.run(["ldd", os.path.join(ROOT, "host", "libse3d_host.so")], capture_output=True, text=True).stdout
    assert "librccl" not in deps   # bound at run time only
