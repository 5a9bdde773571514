"""The C ABI's argument-validation and layout paths under AddressSanitizer -- a HOST-ONLY build (no device code, no GPU; GPU ASan is
not available on this pool and is never attempted): reni_amd/csrc/build_asan_host.sh compiles the library's own translation units
with `hipcc --cuda-host-only -fsanitize=address` and links them with tests/capi/capi_args.c, a plain-C driver that walks every entry
point up to where it would touch the device."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SCRIPT = os.path.join(ROOT, "reni_amd", "csrc", "build_asan_host.sh")


@pytest.mark.skipif(shutil.which("hipcc") is None or not os.path.exists(SCRIPT),
                    reason="needs hipcc and the host build script (.gpurunignore keeps the script off the GPU box: sanitizers run on the CPU build only)")
def test_c_abi_argument_paths_under_asan():
    build = subprocess.run([SCRIPT], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-4000:]
    exe = os.path.join(ROOT, "reni_amd", "csrc", "_build", "asan", "capi_args")
    # (leaks: the HIP runtime's own start-up allocations, not ours -- LeakSanitizer is off, everything else is on)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "AddressSanitizer" not in run.stderr, run.stdout[-2000:] + run.stderr[-6000:]
    assert "checks ok" in run.stdout and int(run.stdout.split()[1]) > 20000
