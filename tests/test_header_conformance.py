"""The drop-in against the REFERENCE's own declarations: scip-sdp_amd/src/sdpi/sdpisolver_hip.c and lapack_interface_hip.c are
compiled (syntax only) with -DHIPSDP_WITH_SCIP, which makes them include /root/reference/src/sdpi/sdpisolver.h (53 prototypes,
:79-724) and lapack_interface.h (7 prototypes, :50-127) instead of the stand-alone headers of include/.  A definition that
disagrees with the reference prototype in any argument is a hard compiler error ("conflicting types").  SCIP's own headers are
not in this image: tests/scip_stubs/ holds test-only stand-ins for the handful of types those two headers need.  Skipped where the
reference tree is absent (the GPU box)."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "sdpi", "sdpisolver.h")), reason="reference tree not present")
@pytest.mark.parametrize("src", ["sdpisolver_hip.c", "lapack_interface_hip.c"])
def test_drop_in_compiles_against_the_reference_headers(src):
    cmd = ["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-DHIPSDP_WITH_SCIP",
           "-I" + os.path.join(ROOT, "tests", "scip_stubs"), "-I" + REF, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "scip-sdp_amd", "src", "sdpi", src)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "conflicting types" not in r.stdout and "warning" not in r.stdout, r.stdout
