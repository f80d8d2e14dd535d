"""The CPU oracle under AddressSanitizer + UBSan (the reference has no race
or memory checking; the restatement gets the compiler's).  CPU only; runs the oracle on a few
golden configurations in a subprocess with the sanitizer runtime preloaded."""
import os
import subprocess
import sys
import textwrap

import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_runs_clean_under_asan_ubsan():
    try:
        so = ol.build(asan=True)
    except subprocess.CalledProcessError:
        pytest.skip("sanitizer build unavailable")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(libasan):
        pytest.skip("libasan not found")
    code = textwrap.dedent(f'''
        import sys, ctypes as C, numpy as np
        sys.path.insert(0, {HERE!r})
        import oracle_lib as ol
        ol._lib = None
        ol.build = lambda *a, **k: {so!r}
        import test_oracle_golden as t
        for name in ("std3_fixed", "funnel_R2P", "dense8_D", "funnel_D_blowup", "corr_R2P_warm", "std3_Y"):
            c, ref = t.load_case(name)
            out = t.run_oracle(c, ref, tape=True)
            assert out["ncalls"] == len(ref["trace"]), name
        print("SANITIZED-OK")
    ''')
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
