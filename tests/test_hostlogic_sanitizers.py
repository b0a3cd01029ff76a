"""The library's HIP-free host logic (gr4-packet-modem_amd/csrc/hostlogic/*.hpp -- the headers the .hip files include:
SyncwordDetectionFilter's gate, SymbolFilter's tag replay / run table, the PayloadMetadataInsert / SyncwordRemove /
HeaderPayloadSplit state machines, the receivers' slot rings and stage loop) built with g++ and no HIP under
AddressSanitizer, UndefinedBehaviorSanitizer and ThreadSanitizer (tests/hostlogic/Makefile: `make SAN=...`, mirroring
the reference's CMakeLists.txt:8-10,81-100) and checked against the CPU oracle on randomised streams.  No GPU."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostlogic")


@pytest.mark.parametrize("san", ["address", "undefined", "thread"])
def test_hostlogic_under_sanitizer(san):
    subprocess.run(["make", "-s", "-C", HERE, f"SAN={san}"], check=True, capture_output=True, text=True)
    exe = os.path.join(HERE, f"hostlogic_san.{san}.bin")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    for seed in ("4", "99"):
        r = subprocess.run([exe, "40", seed], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
        assert "0 failures" in r.stdout
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
