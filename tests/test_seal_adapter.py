"""include/pasta_seal_gfx950_seal.hpp carries pasta::SEALZpCipher / pasta::PASTA_SEAL / sealhelper with the reference's exact
signatures on seal:: types.  Here it is type-checked (g++ -fsyntax-only) against the reference's SEAL 4.0.0 HEADERS together
with a replay of the calls CSP.cpp:238-278 and :296-316 make.  Headers only: the prebuilt libseal-4.0.a is never linked,
loaded or run.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "libs/seal/include/SEAL-4.0")), reason="reference headers not present")
def test_seal_typed_adapter_type_checks_against_the_reference_headers():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-variable",
           "-I" + os.path.join(REF, "libs/seal/include/SEAL-4.0"), "-I" + os.path.join(REF, "src/pasta"),
           "-I" + os.path.join(REF, "libs/keccak"), "-I" + os.path.join(REF, "libs/keccak/opt64"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/cpp/seal_adapter_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
