"""CPU: the host shim under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md §5 "race detection / sanitizers";
round-5 verdict, item 7: lbl_api.hip is 2,500 lines of pools, caches, registries, views and schedule builders behind raw
pointers and no sanitizer had ever seen it).

tests/host_shim/ builds pyrad_amd/csrc/lbl_api.hip - unchanged, as plain C++ - against a stand-in HIP runtime whose
"device" memory is host memory (mock/hip/hip_runtime.h) and launchers that read and write exactly the index ranges the
kernels read and write (mock_kernels.cpp, sharing the launch-shape arithmetic of pyrad_amd/csrc/lbl_launch_shapes.h with the
real launchers), with -fsanitize=address,undefined and scratch blocks without slack (-DLBL_SANITIZER_BUILD).  shim_driver
then goes through the PUBLIC C ABI with seeded random cells, shards, columns, 70-line-list layers, views, option
settings, graph captures, bad arguments, injected allocation failures and random destruction orders; every object the
stand-in runtime handed out must have come back at the end.  CPU only: nothing of this is ever linked into
libpyrad_hip.so or run on the GPU box."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO

DIR = os.path.join(REPO, "tests", "host_shim")
BIN = os.path.join(DIR, "_build", "shim_driver")


@pytest.fixture(scope="module")
def driver():
    if shutil.which("g++") is None:
        pytest.skip("no g++ for the sanitizer build")
    p = subprocess.run(["make", "-C", DIR], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return BIN


def run(driver, seed, rounds, env_extra=None):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.update(env_extra or {})
    return subprocess.run([driver, str(seed), str(rounds)], capture_output=True, text=True, env=env, timeout=600)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_abi_sequences_are_clean_under_asan_and_ubsan(driver, seed):
    p = run(driver, seed, 12)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-6000:])
    assert "no sanitizer report, nothing leaked" in p.stdout and "ERROR" not in p.stderr and "runtime error" not in p.stderr


def test_the_harness_sees_a_block_sized_too_small(driver):
    """self-test: the line-prep stand-in made to write counters past what the shim reserved for them is reported"""
    p = run(driver, 1, 3, {"SHIM_INJECT_OVERRUN": "1"})
    assert p.returncode != 0 and "AddressSanitizer" in p.stderr and "buffer-overflow" in p.stderr
