"""CPU: the sanitizer build of the CPU side (reference toggle common/platform.props:22) --
`make -C oracle sanitize` compiles the oracle and the product's host-only protocol code with
-fsanitize=address,undefined and runs their drivers; any report aborts with a non-zero exit code."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_protocol_are_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitize_check ok" in r.stdout and "filter_sanitize ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
