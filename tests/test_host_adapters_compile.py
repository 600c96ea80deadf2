"""CPU: the host-side C++ that a maintainer drops into OKVIS2 compiles against the interfaces it
claims to implement -- cv::FeatureDetector / cv::DescriptorExtractor subclasses
(okvis2_amd/host/okvfe_opencv_adapters.hpp) and an okvis::ViFrontendInterface subclass
(okvis2_amd/host/okvfe_okvis_frontend.hpp).  OpenCV and OKVIS2 are absent here, so the check runs
against the minimal declarations of tests/mock/ (stand-ins for THIS check only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapters_type_check_against_interface_declarations():
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "mock"),
           os.path.join(ROOT, "tests", "cpp", "adapters_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_vi_frontend_overrides_every_pure_virtual():
    """A class with an un-overridden pure virtual cannot be instantiated: adapters_check.cpp does
    `new okvfe::HipViFrontend(...)`, so the syntax check above fails if a virtual is missed.  Here:
    the header really declares the three overrides with the reference's signatures."""
    src = open(os.path.join(ROOT, "okvis2_amd", "host", "okvfe_okvis_frontend.hpp")).read()
    import re
    for name in ("detectAndDescribe", "dataAssociationAndInitialization", "propagation"):
        assert re.search(r"bool " + name + r"\([^{;]*\)\s*(const\s*)?override\s*\{", src), name
    assert "public okvis::ViFrontendInterface" in src


def test_fp64_reference_dump_tool_type_checks():
    """tools/ref_compare/ref_dump_fp64.cpp (triangulateFast / backProject / matchStereo rows out of real
    Eigen) against declarations of the okvis / Eigen names it uses."""
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "mock", "fp64"),
           os.path.join(ROOT, "tools", "ref_compare", "ref_dump_fp64.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
